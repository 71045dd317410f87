// What the two-launch batch-norm kernels (norm.hip) and the contraction driver (igemm.hip) share: a batch-norm
// reduction over the rows of a contraction's output can ride on the launch that finishes that output (the split-K
// slab reduction) instead of reading the finished tensor again in a launch of its own.
#pragma once
#include "common.h"

namespace gan {

// Decomposition of a [groups][rows][c] tensor shared by the statistics launch and the apply launch of a batch norm:
// (row block) x (tile of 64 channels), at most 256 workgroups.
struct TileBn {
    int groups, rows, c;
    int nb;    // row blocks per group
    int rpb;   // rows per block
};

// Where the rows a batch-norm reduction runs over come from.
struct RowSource {
    const float* slabs;   // nullptr: the rows are read from the finished tensor; else `splits` slabs of `slab_elems` floats
    float* out;           // slabs: every summed row is stored here as well
    const float* bias;    // slabs: per-channel bias added after the slab sum (may be nullptr)
    long slab_elems;
    int splits;
};

// A batch-norm reduction over the output of the NEXT contraction this thread runs (taken and cleared by run_igemm).
struct BnRide {
    int kind;             // 0 none, 1 forward statistics {mean, m2} of the output, 2 backward sums {sum dz, sum dz*xhat}
    float2* partial;      // [groups][nb][c]
    TileBn s;
    // kind 2: the batch norm whose output gradient the contraction produces
    const float* x; const float* gamma; const float* beta; const float* mean; const float* rstd;
    int act; float act_param;
};

BnRide& pending_bn_ride();   // thread-local

// The plan gan_bn_fwd_train (backward = false) / gan_bn_bwd (true) uses for this shape when it takes the two-launch
// path, or false (other paths serve the shape: no ride).
bool bn_ride_plan(int groups, int rows, int c, bool backward, TileBn* s);

// One launch: the ride's reduction over rows taken from `src` (which also finishes the output when it names slabs).
int launch_bn_ride(const BnRide& r, const RowSource& src, hipStream_t st, const char* what);

}  // namespace gan
