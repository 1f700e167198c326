// rt_aux_kernels.h — the small kernels beside render_kernel: tile order (two) and the u8 quantiser.
// Included by rt_amd.hip only (render_kernel's instantiations are compiled group by group in rt_kernels.hip).
#pragma once

#include "rt_kernel.h"

namespace rtk {

// ---------------------------------------------------------------------------------------------
// Tile order.  Tiles differ in cost by an order of magnitude (a tile of deep glass paths casts 40 rays a pixel,
// a tile of matte floor 2), one wave renders one tile, and the frame is over when the last wave is: an expensive
// tile that starts late IS the tail of the kernel.  With the tiles in descending order of cost C4 4K ran 10.5 ->
// 8.7 ms and cover_image 1.54 -> 1.31 ms (profiles/r02_tile_order.txt); an order guessed from the material a
// tile's camera rays hit gained a third of that on C4, nothing on cover_image, and lost on other scenes.  So the
// cost is MEASURED: every wave leaves the number of rounds its path loop took (the deepest pixel of the tile sets
// it; each round is one trace + one shading pass of the whole wave) in a per-tile table of its scene, one store
// per wave, and the next frame of that scene with the same tiling deals its tiles out by that table, most
// expensive first — exact for a camera that stands still (the bench, a turntable's held frame), a good estimate
// for one that moves a little; the first frame of a scene or of a new frame size runs in raster order.
// (Not clock ticks: s_memtime at the top of the kernel counts as a possible write to memory, like the asm of
// profiles/r02_exact_chain_smem_regression.txt, and turned every scalar record load into a vector load —
// cover_image 1.5 -> 2.4 ms.)
// (Area lights: + the points x 64-cell chunks that needed real shadow rays, a round counting as ten of those.)
// Sorting: 32 classes of 1.6 rounds each; rank within a class = arrival (counted per workgroup in LDS, one
// global atomic per class and workgroup — one per tile on the same words serialises in the L2).
// order_buf: [0..31] counters per class, [32 .. 32+n) per tile: class << 27 | arrival index within its class.
constexpr uint32_t kOrderClasses = 32;
__global__ __launch_bounds__(1024) void tile_class_kernel(uint32_t n_tiles, const uint32_t* cost, uint32_t* order_buf) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ uint32_t s_count[kOrderClasses], s_base[kOrderClasses];
    if (threadIdx.x < kOrderClasses) s_count[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t cls = 0, rank = 0;
    if (tile < n_tiles) {
        const uint32_t b = cost[tile] >> 4;
        cls = b > 31u ? 31u : b;
        rank = atomicAdd(&s_count[cls], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kOrderClasses) s_base[threadIdx.x] = s_count[threadIdx.x] ? atomicAdd(order_buf + threadIdx.x, s_count[threadIdx.x]) : 0u;
    __syncthreads();
    if (tile < n_tiles) order_buf[kOrderClasses + tile] = (cls << 27) | (s_base[cls] + rank);
}
// order[position] = tile, the highest class first
__global__ void tile_order_kernel(uint32_t n_tiles, const uint32_t* order_buf, uint32_t* order) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t v = order_buf[kOrderClasses + tile], cls = v >> 27, at = v & 0x07FFFFFFu;
    uint32_t base = 0;
    for (uint32_t c = kOrderClasses - 1u; c > cls; --c) base += order_buf[c];
    order[base + at] = tile;
}

__global__ void quantize_kernel(const float* __restrict__ rgb, size_t n, uint8_t* __restrict__ out) {
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) out[i] = quantize_u8(rgb[i]);  // canvas.rs:37-39
}

// The eight direction-ordered LNode streams (rt_math.h lane_place) from the canonical one: one thread per record walks
// down from the root to its own record (the tree's depth in steps; `guard` in lane_place ends a walk whatever the links
// say) and stores the record at its place in each order, with that order's skip link.
__global__ void lane_streams_kernel(const LNode* __restrict__ canon, uint32_t n, uint32_t n_top, const uint32_t* __restrict__ portal_ret,
                                    LNode* __restrict__ streams) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lane_place(canon, n, n_top, portal_ret, i, streams);
}

}  // namespace rtk
