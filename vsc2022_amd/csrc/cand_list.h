// The pre-filters' candidate list: one private segment per wave (no atomics) + a shared tail for the waves whose
// segment is full.  The tail is handed out in CHUNKS -- one atomic per `tail_chunk` entries instead of one per
// ballot group: with skewed candidate distributions (score-normalised descriptors: a few percent of the query rows
// own most of the hits) a third of a launch's candidates can take this route, and 2048 waves queueing on one
// counter once per handful of candidates turned 40 ms launches into seconds.  A wave keeps its current chunk as a
// private extension of its segment; what it leaves unused is marked i = -1, which rescore_list skips.
#pragma once
#include "vscmi_common.h"

namespace vscmi {

// The wave's tail state lives in LDS, not in registers: it is touched only on the rare "segment full" path, and the
// pre-filter kernels have no register to spare.  The code below is inlined 128 times into fully unrolled loops over
// the accumulator registers: it must stay loop-free -- with a loop inside, the compiler stopped unrolling the outer
// loops, indexed the accumulators dynamically and moved all 128 of them to scratch memory after every tile (both
// panel kernels at 40 % of their rate; measured, round 3).  Hence no sentinel fill: every chunk has a fill level,
// tail_fill[chunk id], written by its owner when it moves on (or at the end of the kernel); rescore_list reads it.
struct TailExt {
    long long pos;  // next free entry of the wave's current chunk (absolute index into the candidate arrays)
    int left;       // free entries in it; < 0: the tail has overflowed, the launch is lost
    int have;       // 1: the wave owns a chunk
};

__device__ __forceinline__ void tail_init(TailExt* e, int ln) {
    if (ln == 0) { e->pos = 0; e->left = 0; e->have = 0; }
}

// Room for `total` (<= 64) more candidates of this wave behind its full segment: true and `pos` = first entry, or
// false (tail exhausted: bit 1 of *overflow is set -- bit 0 belongs to the kept-hit list -- and the host reruns the
// search on the fp16 kernel or with larger buffers).  Wave-uniform arguments; ln = lane;
// `e` = the wave's slot in LDS (one wave reads and writes it, in program order); chunk = 1 << chunk_shift.
__device__ __forceinline__ bool tail_take(unsigned long long* tail_count, long long tail_cap, long long tail_base,
                                          int chunk_shift, int* tail_fill, int* overflow, int total, int ln, TailExt* e,
                                          int64_t& pos) {
    const int chunk = 1 << chunk_shift;
    int left = e->left;
    long long p = e->pos;
    if (left < 0) return false;
    if (total > chunk) {  // (only the block-at-a-time emitter asks for more than 64 at once: rerun with larger buffers)
        if (ln == 0) { atomicOr(overflow, 2); e->left = -1; }
        return false;
    }
    if (total > left) {
        // close the current chunk (its fill level), take the next one
        if (ln == 0 && e->have) tail_fill[(p - 1 - tail_base) >> chunk_shift] = chunk - left;
        unsigned long long base = 0;
        if (ln == 0) base = atomicAdd(tail_count, (unsigned long long)chunk);
        base = __shfl(base, 0);
        if ((long long)(base + chunk) > tail_cap) {
            if (ln == 0) { atomicOr(overflow, 2); e->left = -1; e->have = 0; }
            return false;
        }
        p = tail_base + (long long)base;
        left = chunk;
    }
    pos = p;
    if (ln == 0) { e->pos = p + total; e->left = left - total; e->have = 1; }
    return true;
}

// at the end of the kernel: the fill level of the wave's last chunk
__device__ __forceinline__ void tail_close(long long tail_base, int chunk_shift, int* tail_fill, int ln, TailExt* e) {
    if (ln == 0 && e->have && e->left >= 0) tail_fill[(e->pos - 1 - tail_base) >> chunk_shift] = (1 << chunk_shift) - e->left;
}

// chunk size of a launch (a power of two, as a shift): large enough that the atomics vanish, small enough that every
// wave can leave one chunk half empty without exhausting the tail
inline int tail_chunk_shift_for(long long tail_cap, int n_seg) {
    long long c = tail_cap / (8ll * (n_seg > 0 ? n_seg : 1));
    int sh = 6;  // 64 entries
    while (sh < 12 && (2ll << sh) <= c) ++sh;
    return sh;
}

}  // namespace vscmi
