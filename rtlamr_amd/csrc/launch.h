// Kernel launchers, one translation unit per kernel family so that the library builds in parallel (make -j): the
// template instantiations (ten chip lengths of K1, 40 variants of the walk search) are what takes the compile time.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "k1_common.h"
#include "k2_common.h"

namespace amr {

// K1 for one chip length (flags.go:127-132), whole wave-tiles of 64 blocks; false: not a legal chip length
bool launch_k1(int chip_length, dim3 grid, hipStream_t st, const K1Args &a, hipEvent_t start, hipEvent_t stop);
// (grid.x == 0, a.qt == nullptr: no launch -- the name of the kernel this chip length gets, as rocprofv3 prints it, into the
// host buffer a.iq points to, a.n_blocks bytes)
// K1 for the blocks behind the last whole wave-tile (and for the single block of the unchanged main.go loop): one wave per
// block (k1_coop.h), blocks first_block .. first_block + n_blocks - 1 of the launch
bool launch_k1_coop(int chip_length, uint32_t first_block, uint32_t n_blocks, hipStream_t st, K1Args a, hipEvent_t start, hipEvent_t stop);

// K2, the walk search (k2_walk.h) for the set of rtlamr's preambles given as a mask (bit k = kind k of k2_walk_kind_of).
// false: no kernel for this SymbolLength (nothing launched)
bool launch_k2_walk(uint32_t symbol_length, uint32_t set, uint32_t grid, size_t lds_bytes, hipStream_t st, hipEvent_t start, hipEvent_t stop,
                    const K2Args &a, hipError_t *err);
// K2 for ONE of rtlamr's preambles (kind 0..3 of k2_walk_kind_of) with the whole row in registers (k2_row.h): every
// single-preamble decoder whose rows have 16..256 words.  extra_wgs: workgroups behind the searching ones (state update,
// deferred-block copies).  false: no kernel for this (SymbolLength, kind, row length)
bool launch_k2_row(uint32_t symbol_length, uint32_t kind, uint32_t extra_wgs, size_t lds_bytes, hipStream_t st, hipEvent_t start, hipEvent_t stop,
                   const K2Args &a, hipError_t *err);
// the clean-up launch behind an in-wave search (k1_search.h: K1 searched every tile but row 63's last words): rows of 16 words,
// kind 0 / 1.  a.qt == nullptr: no launch, the answer says whether the geometry has one.  false: none
bool launch_k2_cleanup(uint32_t symbol_length, uint32_t kind, hipStream_t st, hipEvent_t start, hipEvent_t stop, const K2Args &a);
// which of rtlamr's four preambles (k2_walk.h) a registered preamble is, -1: none
int k2_walk_kind_of(uint32_t len, uint64_t bits);
// K2 fallbacks (k2_search.h): list-based for up to four short preambles / short rows, dense for everything else
bool launch_k2_fast(uint32_t n_pre, int nwv, uint32_t grid, size_t lds_bytes, hipStream_t st, hipEvent_t start, hipEvent_t stop,
                    const K2Args &a, hipError_t *err);
void launch_k2_dense(uint32_t grid, size_t lds_bytes, hipStream_t st, hipEvent_t start, hipEvent_t stop, const K2Args &a,
                     hipError_t *err);

}  // namespace amr
