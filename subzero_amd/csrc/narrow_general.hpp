// narrow_general.hpp -- launchers of the lane-per-pair and wave-per-pair narrow-phase kernels (narrow_general.hip): the
// single-minimum fast path over what the team pass hands on, the general Vatti sweep for what that defers, the floes at
// the walls.  A translation unit of their own: they instantiate the whole of vatti.hpp / contact.hpp and take minutes to
// compile, the kernels of step.hip seconds.
#pragma once
#include "step_state.hpp"

#define CONVEX_STRIDE 616
#define CONVEX_BIG_STRIDE 1128
#define FAST_WS_STRIDE 2344

int sz_narrow_general_configure(szContext *ctx, bool narrow_in_lds);
void szl_k_narrow(dim3 grid, size_t lds, hipStream_t st, i64 first, i64 count, const int *list, int mult, int lanes, int lds_per_lane, const Floes &F, const Params &P, const Work &W);
void szl_k_narrow_lds(dim3 grid, size_t lds, hipStream_t st, i64 count, const int *list, int mult, int lanes, int lds_per_lane, const Floes &F, const Params &P, const Work &W);
void szl_k_narrow_wave(dim3 grid, size_t lds, hipStream_t st, i64 count, const int *list, int lds_per_wave, int tasks, const Floes &F, const Params &P, const Work &W);
void szl_k_narrow_walls(dim3 grid, size_t lds, hipStream_t st, int lds_per_wave, int tasks, const Floes &F, const Params &P, const Work &W);
void szl_k_narrow_convex(dim3 grid, size_t lds, hipStream_t st, i64 first, i64 count, const unsigned long long *dev_count, int lanes, const Floes &F, const Params &P, const Work &W, const int *order);
void szl_k_narrow_convex_big(dim3 grid, size_t lds, hipStream_t st, i64 count, const int *list, int lanes, const Floes &F, const Params &P, const Work &W);
void szl_k_narrow_a(dim3 grid, size_t lds, hipStream_t st, i64 first, i64 count, int lanes, const Floes &F, const Params &P, const Work &W);
void szl_k_narrow_b(dim3 grid, size_t lds, hipStream_t st, int lanes, const Floes &F, const Params &P, const Work &W);
