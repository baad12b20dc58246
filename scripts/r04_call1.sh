#!/bin/bash
# round 4, call 1: the per-read overflow fix on the GPU (sliced tests), the box's topology, and what its tmpfs takes as a sink
mkdir -p gpurun_out
{
  uname -a; nproc; lscpu | head -40
  for f in /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/shmem_enabled /sys/kernel/mm/transparent_hugepage/defrag; do echo "$f: $(cat $f 2>/dev/null)"; done
  for n in /sys/devices/system/node/node*; do echo "$n cpulist $(cat $n/cpulist) $(grep MemTotal $n/meminfo)"; done
  cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list
  df -h /dev/shm /tmp
  free -g
  ls /sys/class/drm/ 2>/dev/null; for d in /sys/class/drm/card*/device/numa_node; do echo "$d $(cat $d)"; done
  rocm-smi --showtoponuma 2>/dev/null | head -20
} > gpurun_out/r04_box_topology.log 2>&1
g++ -O2 -pthread -o /tmp/sink_probe scripts/r04_sink_probe.cpp && /tmp/sink_probe /dev/shm > gpurun_out/r04_sink_probe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_sliced.py -x -q -m gpu > gpurun_out/r04_sliced_tests.log 2>&1
tail -3 gpurun_out/r04_sliced_tests.log
cat gpurun_out/r04_sink_probe.log
