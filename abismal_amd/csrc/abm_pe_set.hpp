// abismal_amd: the paired-end candidate set (pe_candidates, src/abismal.cpp:775-863) of one wave -- a header of
// its own so that tests/hip/pe_set_check.hip can drive it against libstdc++ without compiling the mapping kernels.
#pragma once
#include "abm_kernels_core.hpp"

namespace abm {

struct PeLds {
  u32 *heap;     // [cap] live set as a binary heap of diffs<<16 | handle; reused as sort buffer
  u32 *lpos[2];  // finished sets of the two ends of this orientation call: positions
  i16 *ld[2];    //   ... diffs
  i16 *lsc[2];   //   ... alignment scores of the entries that can pair up
  u32 *jidx;     // [kSeCap] which list entry an alignment job belongs to
  u32 *tmp;      // [cap] scratch table (tier 2: global memory)
  u32 cap;
};

// list element reads (tier 1: LDS; tier 2: this wave's lists in global memory, through L1 -- the
// binary searches of the mating code revisit the same few lines from every lane)
template <bool BIG> __device__ __forceinline__ u32 ld_list(const u32 *p) { return *p; }
template <bool BIG> __device__ __forceinline__ int ld_list(const i16 *p) { return static_cast<int>(*p); }

// pe_candidates, src/abismal.cpp:775-863, one set per wave.
//
// What the reference's heap does depends on the pass (:824-842):
//  * specific pass (cutoff <= good_cutoff throughout, because set_specific starts it there):
//    every admitted hit has diffs <= good_cutoff, so a full set GROWS by one instead of
//    evicting, until 32768 entries.  The sentinel {0.4 L, pos 0} is never evicted, the cutoff
//    never moves, and the set is simply "every hit within good_cutoff, in stream order".
//    Nothing reads the heap's array order here, so hits are APPENDED, a whole chunk of 64
//    candidates per step, straight into the list the mating code will sort.
//  * sensitive pass: only runs if the set never grew (capacity still 32, :793-797); the cutoff
//    restarts at the heap's top, hits beyond good_cutoff evict the top of a full set and the
//    others still grow it (check_hits passes specific=true in both passes, :1364-1369).
//  * a set that reaches 32768 entries starts evicting in the specific pass as well.
// The last two need the real heap; it is rebuilt on demand by replaying std::push_heap over
// the appended entries in their order of arrival (heapify), which reproduces libstdc++'s array
// exactly.  Heap entries are (diffs << 16 | handle) keys, handle = slot in the position list.
//
// std::push_heap / std::pop_heap are restated so that the whole wave works on one operation:
// a sift touches only one root-to-leaf path, so the lanes fetch the path (push: the hole's
// ancestors, one per lane; pop: the 63 nodes of six heap levels at a time) in one memory round
// trip, the walk itself runs on registers, and the moved keys are stored in parallel.
struct PeSet {
  static constexpr bool kFifo = true;
  static constexpr bool kAppend = true;
  u32 *heap;   // [cap] keys, valid only while `heaped`
  u32 *lpos;   // [cap] positions: arrival order while appending, handle-indexed once heaped
  i16 *ld;     // [cap] diffs of the same entries (not maintained once heaped: the keys hold them)
  u32 cap_avail;
  // the seed kernel of the phase-split launches: a list still growing by appends that outgrows its LDS slot moves to the
  // wave's staging area in global memory (spill_cap entries; 0 = none) and carries on there -- appended entries are
  // written once and not read again before the list is handed over, so where they lie costs nothing
  u32 *spill_pos;
  i16 *spill_d;
  u32 spill_cap;
  bool spilled;
  u32 top;     // heap[0]
  int sz, capacity, cutoff, good_cutoff;
  bool sure_ambig, overflow, heaped;

  __device__ __forceinline__ static int key_d(u32 k) { return static_cast<int>(k) >> 16; }
  __device__ __forceinline__ void begin_read(u32 readlen) {
    const int worst = static_cast<i16>(0.4 * readlen);
    top = static_cast<u32>(worst) << 16;  // sentinel, handle 0 -> pos 0
    if (lane_id() == 0) { lpos[0] = 0; ld[0] = static_cast<i16>(worst); }
    wave_sync();
    sz = 1;
    capacity = static_cast<int>(kPeCapSmall);
    cutoff = worst;
    good_cutoff = static_cast<i16>(readlen / 10);
    sure_ambig = false;
    overflow = false;
    heaped = false;
  }
  __device__ __forceinline__ bool wants_sensitive() const {
    return capacity == static_cast<int>(kPeCapSmall) || cutoff > good_cutoff;
  }
  // specific pass, not heaped: the chunk's survivors (lanes in `todo`, distance h, position pos) all
  // go in, in lane order.  Returns the lanes still to be offered one by one (none, unless the set
  // just reached 32768 entries and turned into a heap).
  __device__ __forceinline__ u64 append(u64 todo, int h, u32 pos) {
    const int lane = lane_id();
    const int cnt = __popcll(todo);
    if (spill_cap > cap_avail && sz + cnt > static_cast<int>(cap_avail)) {  // (only ever true in the seed kernel, once per list)
      for (int i = lane; i < sz; i += 64) { spill_pos[i] = lpos[i]; spill_d[i] = ld[i]; }
      lpos = spill_pos; ld = spill_d;
      cap_avail = spill_cap;
      spill_cap = 0;
      spilled = true;
      wave_sync();
    }
    const int limit = static_cast<int>(min(cap_avail, kPeCapLarge));
    const int take = min(cnt, limit - sz);
    const int rank = __popcll(todo & ((1ull << lane) - 1));
    const bool mine = (todo >> lane) & 1ull;
    if (mine && rank < take) { lpos[sz + rank] = pos; ld[sz + rank] = static_cast<i16>(h); }
    sz += take;
    capacity = max(capacity, sz);
    sure_ambig = (sz == capacity) && cutoff == 0;
    if (take == cnt) return 0;
    if (cap_avail < kPeCapLarge) {  // this tier ran out of room: redo in the next
      overflow = true;
      sure_ambig = true;
      return 0;
    }
    wave_sync();
    heapify();
    return todo & __ballot(mine && rank >= take);
  }
  // the heap std::push_heap leaves after the entries arrived one by one
  __device__ __forceinline__ void heapify() {
    const int lane = lane_id();
    for (int i0 = 0; i0 < sz; i0 += 64) {
      const int dl = i0 + lane < sz ? static_cast<int>(ld[i0 + lane]) : 0;
      const int m = min(64, sz - i0);
      for (int k = 0; k < m; ++k)
        push(i0 + k, (static_cast<u32>(rdlane(dl, k)) << 16) | static_cast<u32>(i0 + k));
    }
    heaped = true;
  }
  __device__ __forceinline__ void set_sensitive() {
    if (!heaped) heapify();
    cutoff = key_d(top);
  }
  // std::push_heap of `key` into slot `hole` (libstdc++ __push_heap): lane t holds the hole's
  // t-th ancestor; parents with fewer diffs than the key move down one level
  __device__ __forceinline__ void push(int hole, u32 key) {
    const int lane = lane_id();
    const int node = lane < 17 ? ((hole + 1) >> lane) - 1 : -1;
    const int par = lane < 17 ? ((hole + 1) >> (lane + 1)) - 1 : -1;
    u32 pk = 0;
    if (par >= 0) pk = heap[par];
    const bool moves = par >= 0 && key_d(pk) < key_d(key);
    const int stop = __builtin_ctzll(~__ballot(moves));
    if (lane < stop) heap[node] = pk;
    else if (lane == stop) heap[node] = key;
    if (((hole + 1) >> stop) == 1) top = key;
    wave_sync();
  }
  // (Round 5 tried keeping the heap's first ten levels in LDS -- 4 KB the alignments leave idle during the seed passes -- so
  // that the walk down needs one trip to memory instead of three: the costliest pairs, whose 32768-entry sets see a couple
  // of hundred thousand updates, got 20 % SLOWER (replay 1.9 -> 2.3 G cycles for the worst one, the tier-2 launch alone
  // 1.10-1.14 -> 1.32 s; profiles/r05_exp_heap_levels_in_lds.log): a lone wave is bound by the instructions of an update,
  // not by its round trips, and two address spaces per access add to them.)
  // std::pop_heap on [0,n) (libstdc++ __adjust_heap: the hole follows the larger child -- the
  // right one on ties -- down to a leaf, then the last element sifts up from there); returns the
  // handle of the evicted top.  Slot n-1 is left for the caller to refill.
  __device__ __forceinline__ int pop_max(int n) {
    const int lane = lane_id();
    const int len = n - 1;
    const int freed = static_cast<int>(top & 0x7FFFu);
    // this lane's place in a six-level subtree: level lt, j-th node of that level (lane 63: the last element)
    const int lt = 31 - __clz(lane + 1);
    const int lj = lane + 1 - (1 << lt);
    int g = 0, gdepth = 0;
    u32 K;
    auto load_subtree = [&](int root, int depth) {
      g = root; gdepth = depth;
      const int idx = lane == 63 ? len : ((root + 1) << lt) - 1 + lj;
      K = idx <= len ? heap[idx] : 0u;
    };
    auto key_at = [&](int x, int t) { return rdlane(K, (1 << t) - 1 + (x - (((g + 1) << t) - 1))); };
    load_subtree(0, 0);
    const u32 vk = rdlane(K, 63);
    int hole = 0, second = 0, depth = 0;
    int pidx = 0;   // lane t: the path's node at depth t ...
    u32 pnew = 0;   // ... and the child key that moves into it
    while (second < (len - 1) / 2) {
      second = 2 * (second + 1);
      int t = depth + 1 - gdepth;
      if (t > 5) { load_subtree(hole, depth); t = 1; }
      u32 sk = key_at(second, t);
      const u32 lk = key_at(second - 1, t);
      if (key_d(sk) < key_d(lk)) { --second; sk = lk; }
      if (lane == depth) { pidx = hole; pnew = sk; }
      hole = second;
      ++depth;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
      second = 2 * (second + 1);
      int t = depth + 1 - gdepth;
      if (t > 5) { load_subtree(hole, depth); t = 1; }
      const u32 lk = key_at(second - 1, t);
      if (lane == depth) { pidx = hole; pnew = lk; }
      hole = second - 1;
      ++depth;
    }
    if (lane == depth) pidx = hole;
    // __push_heap of the last element from the leaf: the path's nodes now hold their children's
    // keys; while the one above has fewer diffs than vk it moves back down (restoring the old key)
    const u32 above = __shfl_up(pnew, 1);
    const bool back = lane >= 1 && lane <= depth && key_d(above) < key_d(vk);
    const u64 stay = ~__ballot(back) & ((2ull << depth) - 1);
    const int ts = 63 - __builtin_clzll(stay);
    if (lane < ts) heap[pidx] = pnew;
    else if (lane == ts) heap[pidx] = vk;
    top = ts == 0 ? vk : rdlane(pnew, 0);
    wave_sync();
    return freed;
  }
  // A RUN of survivors that tie with the cutoff c while the set is full at 32768 entries and the top, the last
  // element and the last element's parent all sit at c: SeSet::tie_run's shift register (see there for why) on the
  // heap in memory.  The hole's path is walked once, only as far as its nodes hold a c; the chain p0 .. pj, last
  // then lives in lanes 0 .. j + 1, every tie is one lane shift plus a store of the newcomer's position into the
  // evicted payload slot, and the chain is written back once -- instead of four dependent memory round trips per
  // tie (a pair whose ends lie in satellite repeats offers hundreds of thousands of them).
  __device__ __forceinline__ int tie_run(u64 &todo, u64 ties, u32 cand_pos, u32 /*flags*/) {
    static_assert((kPeCapLarge & 1u) == 0, "the hole's path assumes every inner node of the popped heap has two children");
    if (overflow || !heaped || sz != static_cast<int>(kPeCapLarge) || capacity != static_cast<int>(kPeCapLarge)) return 0;
    const int c = cutoff;
    if (key_d(top) != c) return 0;
    const u64 brk = todo & ~ties;
    const u64 run = brk ? (todo & ((brk & (0 - brk)) - 1)) : todo;
    if (run == 0) return 0;
    const int lane = lane_id();
    const int len = sz - 1;
    const int lt = 31 - __clz(lane + 1);
    const int lj = lane + 1 - (1 << lt);
    int g = 0, gdepth = 0;
    u32 K;
    auto load_subtree = [&](int root, int depth) {
      g = root; gdepth = depth;
      const int idx = lane == 63 ? len : ((root + 1) << lt) - 1 + lj;
      K = idx <= len ? heap[idx] : 0u;
    };
    auto key_at = [&](int x, int t) { return rdlane(K, (1 << t) - 1 + (x - (((g + 1) << t) - 1))); };
    const u32 kpar = heap[(len - 1) >> 1];
    load_subtree(0, 0);
    const u32 klast = rdlane(K, 63);
    if (key_d(klast) != c || key_d(static_cast<u32>(uni(static_cast<int>(kpar)))) != c) return 0;
    // the hole's path (as pop_max walks it) while its nodes hold a c; lane t keeps node t of the chain and its key
    int cnode = 0;
    u32 ckey = top;
    int second = 0, depth = 0;
    while (second < (len - 1) / 2) {
      int child = 2 * (second + 1);
      int t = depth + 1 - gdepth;
      if (t > 5) { load_subtree(second, depth); t = 1; }
      u32 sk = key_at(child, t);
      const u32 lk = key_at(child - 1, t);
      if (key_d(sk) < key_d(lk)) { --child; sk = lk; }
      if (key_d(sk) != c) break;
      second = child;
      ++depth;
      if (lane == depth) { cnode = child; ckey = sk; }
    }
    const int n = depth + 2;  // chain length: depth + 1 path nodes and the last element
    if (lane == n - 1) { cnode = len; ckey = klast; }
    const int m = __popcll(run);
    todo &= ~run;
    int play = m;
    while (play >= 2 * n) play -= n;
    u64 last = 0, rest = run;
    for (int k = 0; k < play; ++k) {
      const int hi = 63 - __builtin_clzll(rest);
      last |= 1ull << hi;
      rest &= ~(1ull << hi);
    }
    const u32 c16 = static_cast<u32>(c) << 16;
    while (last) {
      const int l = __builtin_ctzll(last);
      last &= last - 1;
      const u32 slot = rdlane(ckey, 0) & 0x7FFFu;
      const u32 p = rdlane(cand_pos, l);
      const u32 nxt = static_cast<u32>(from_next_lane(static_cast<int>(ckey)));
      ckey = lane < n - 1 ? nxt : (lane == n - 1 ? (c16 | slot) : ckey);
      if (lane == 0) lpos[slot] = p;
    }
    if (lane < n) heap[cnode] = ckey;
    top = rdlane(ckey, 0);
    wave_sync();
    return m;
  }

  // pe_candidates::update, :824-842, on the heap
  __device__ __forceinline__ void admit(bool specific, int d, u32 /*flags*/, u32 p) {
    if (overflow) return;
    if (!heaped) heapify();
    int handle;
    if (sz == capacity) {
      if (specific && capacity != static_cast<int>(kPeCapLarge) && d <= good_cutoff) {
        if (capacity == static_cast<int>(cap_avail)) {  // this tier ran out of room: redo in the next
          overflow = true;
          sure_ambig = true;
          return;
        }
        ++capacity;
        handle = sz;
      }
      else {
        handle = pop_max(sz);
        --sz;
      }
    }
    else
      handle = sz;
    if (lane_id() == 0) lpos[handle] = p;
    ++sz;
    push(sz - 1, (static_cast<u32>(d) << 16) | static_cast<u32>(handle));
    const int topd = key_d(top);
    cutoff = specific ? min(cutoff, topd) : topd;
    sure_ambig = (sz == capacity) && cutoff == 0;
  }
};

}  // namespace abm
