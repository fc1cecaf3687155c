// Frame-deduplicated replay: the record that travels actor -> learner ("run record").
//
// The reference stores both frame stacks of every transition in full (core/memories/shared_memory.py:19-24):
// 2 x 4 x 84 x 84 bytes for what is, after the first step of an episode, ONE new 84 x 84 frame.  Here an actor-side
// packer (shm_ring.cpp, NStep with a RunPacker) keeps each env's frames as a RUN of consecutive planes: a state is a
// window of C consecutive planes of its run, so a state that slides by one frame adds one plane, and a transition is
// (tail floats, offset of s0's window, offset of sN's window).  A run record carries up to `plane_cap` planes and the
// transitions whose two windows lie inside them; the learner copies the planes contiguously into a plane ring in HBM
// and turns the two offsets into plane indices of that ring (ops/csrc/replay_kernels.hip: replay_unpack_runs_kernel),
// which the Nature-CNN kernels read as `planes + ref[idx[b]] * plane_stride` -- the same 28 224 contiguous bytes as
// before, at a stride of one plane.
//
// Record layout (all offsets multiples of 16):
//   [RunHeader 64 B][transition table: trans_cap x trans_stride][planes: plane_cap x plane_stride]
//   transition entry: [tail floats: action | reward | gamma | terminal | priority][int32 off0][int32 offN] padded to 16
#pragma once
#include <cstdint>

#define PDA_RUN_MAGIC 0x52554e31u   // "RUN1"

struct PdaRunHeader {
    uint32_t magic;
    int32_t num_planes;         // planes used (<= plane_cap)
    int32_t num_transitions;    // table entries used (<= trans_cap)
    int32_t window;             // C: planes per state
    int32_t reserved[12];
};
static_assert(sizeof(PdaRunHeader) == 64, "run header is one cache line");

static inline uint32_t pda_run_trans_stride(uint32_t tail_floats) { return (4u * tail_floats + 8u + 15u) / 16u * 16u; }
static inline uint32_t pda_run_plane_stride(uint32_t plane_bytes) { return (plane_bytes + 15u) / 16u * 16u; }
static inline uint64_t pda_run_planes_offset(uint32_t tail_floats, uint32_t trans_cap) {
    return 64ull + (uint64_t)trans_cap * pda_run_trans_stride(tail_floats);
}
static inline uint64_t pda_run_record_bytes(uint32_t tail_floats, uint32_t trans_cap, uint32_t plane_bytes,
                                            uint32_t plane_cap) {
    return pda_run_planes_offset(tail_floats, trans_cap) + (uint64_t)plane_cap * pda_run_plane_stride(plane_bytes);
}
