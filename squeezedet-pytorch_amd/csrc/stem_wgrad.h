// Shared between wgrad.hip (dense stem weight-gradient kernels, entry points) and stem_wgrad_gather.hip.
#pragma once
#include "sqd_common.h"

struct StemWgradArgs {
  const float* dy; const float* img; float* slab;
  const float* pooled; const unsigned char* amax;    // POOLED variant: dy is dPool [B][Hp][Wp][N]
  int Hp, Wp;
  int B, Hin, Win, Ho, Wo, N;
  int tiles_x, tiles_y, nblocks;
  long long slab_stride;
};

// stem_wgrad_gather.hip: the gather form of the pooled stem weight gradient (3x3 / 64-channel stem).  Launches one workgroup per
// slab, at most S; returns the number of slabs written (the caller reduces exactly that many) or -1 on a launch failure.
int launch_stem_wgrad_gather(StemWgradArgs a, int S, hipStream_t s);
