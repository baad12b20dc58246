#!/bin/bash
# round 4, call 19: what a kernel that does not fit the instruction cache costs -- the single-end kernel with its seed passes
# instantiated once per call of a read (49 KB -> 122 KB of code; T-rich mode runs two of the four copies, random PBAT all four)
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r04_call19_index.err
{
for rep in 1 2; do
  for v in base bloat; do
    cp abismal_amd/_ab/libabismal_amd_$v.so abismal_amd/libabismal_amd.so
    python3 bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v rep $rep  T-rich 100 bp: kernel %.1f ms  %.2f M reads/s' % (d['roofline']['avg_kernel_ms'], d['value']/1e6))"
    python3 bench.py --mode random --read-len 150 --reads 4000000 --steps 3 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v rep $rep  random PBAT 150 bp: kernel %.1f ms  %.2f M reads/s' % (d['roofline']['avg_kernel_ms'], d['value']/1e6))"
  done
done
cd /tmp
for v in base bloat; do
  cp $REPO/abismal_amd/_ab/libabismal_amd_$v.so $REPO/abismal_amd/libabismal_amd.so
  rm -rf /tmp/prof_cb
  (cd "$REPO" && timeout 600 rocprofv3 --pmc SQC_TC_INST_REQ TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/prof_cb -- python3 bench.py --mode random --read-len 150 --reads 4000000 --no-cpu-baseline --no-e2e --no-other-configs --no-stage-split --steps 1 --warmup 0 > /tmp/cb.log 2>&1)
  CC=$(find /tmp/prof_cb -name '*counter_collection.csv' | head -1)
  [ -n "$CC" ] && grep "map_se_kernel<false" "$CC" | awk -F, -v v=$v '{print v, $(NF-3), $(NF-2), ($NF-$(NF-1))/1e6 " ms"}'
done
cp $REPO/abismal_amd/_ab/libabismal_amd_base.so $REPO/abismal_amd/libabismal_amd.so
} > gpurun_out/r04_exp_se_code_bloat.log 2>&1
cat gpurun_out/r04_exp_se_code_bloat.log
