// Access-pattern probe: does copying a tensor in the 16x16-tile pattern of the convolution kernels (16 rows x 512 B segments at a 2 KB
// stride per channel-group plane) reach the bandwidth of a linear copy?  Build: hipcc --offload-arch=gfx950 -shared -fPIC -O3.
#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one workgroup = one 16x16 tile of one 64x64 block, all 8 channel-group planes ([n][8][64][64] pixels of 32 B)
template <int HALO>
__global__ __launch_bounds__(256) void tile_copy(const char *__restrict__ in, char *__restrict__ out)
{
    int bid = blockIdx.x;
    bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);          // XCD-aware as in the convolution kernels
    const int n = bid >> 4, t = bid & 15, ty = t >> 2, tx = t & 3, tid = threadIdx.x;
    const size_t plane = 64 * 64 * 32, blk = 8 * plane;
    u32x4 r[16];
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int p = it * 256 + tid, row = p >> 5, col = p & 31;
            r[g * 2 + it] = *reinterpret_cast<const u32x4 *>(in + n * blk + g * plane + (size_t)((ty * 16 + row) * 64 + tx * 16) * 32 + col * 16);
        }
    if (HALO) {      // the two extra rows above / below and a 32-B column left / right, clamped into the block (read only)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int hrow = tid >> 6, c = tid & 63;          // 4 pieces x 64: rows -1, 16 (36 x 16 B each) and the two side columns
            int y, xb;
            if (hrow < 2) { y = hrow ? ty * 16 + 16 : ty * 16 - 1; xb = (tx * 16 - 1) * 32 + (c % 36) * 16; }
            else { y = ty * 16 + (c & 15); xb = (hrow == 2 ? tx * 16 - 1 : tx * 16 + 16) * 32 + (c >> 5) * 16; }
            y = min(max(y, 0), 63); xb = min(max(xb, 0), 64 * 32 - 16);
            const u32x4 h = *reinterpret_cast<const u32x4 *>(in + n * blk + g * plane + (size_t)y * 64 * 32 + xb);
            acc.x ^= h.x; acc.y ^= h.y; acc.z ^= h.z; acc.w ^= h.w;
        }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int p = it * 256 + tid, row = p >> 5, col = p & 31;
            u32x4 v = r[g * 2 + it];
            if (HALO) v.x ^= acc.x & 0;          // keep the halo reads alive
            *reinterpret_cast<u32x4 *>(out + n * blk + g * plane + (size_t)((ty * 16 + row) * 64 + tx * 16) * 32 + col * 16) = v;
        }
    if (HALO && (acc.x | acc.y | acc.z | acc.w) == 0x12345u) out[0] = 1;
}

// the same bytes, every workgroup a contiguous 64 KB
__global__ __launch_bounds__(256) void linear_copy(const char *__restrict__ in, char *__restrict__ out)
{
    int bid = blockIdx.x;
    bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const size_t base = (size_t)bid * 65536;
    u32x4 r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = *reinterpret_cast<const u32x4 *>(in + base + (size_t)(i * 256 + threadIdx.x) * 16);
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x4 *>(out + base + (size_t)(i * 256 + threadIdx.x) * 16) = r[i];
}

// tile-major layout: the 8 KB of a tile's group plane are contiguous ([n][tile 16][8][16x16 px])
__global__ __launch_bounds__(256) void tilemajor_copy(const char *__restrict__ in, char *__restrict__ out)
{
    int bid = blockIdx.x;
    bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const size_t base = (size_t)bid * 65536;
    u32x4 r[16];
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int it = 0; it < 2; ++it) r[g * 2 + it] = *reinterpret_cast<const u32x4 *>(in + base + g * 8192 + (size_t)(it * 256 + threadIdx.x) * 16);
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int it = 0; it < 2; ++it) *reinterpret_cast<u32x4 *>(out + base + g * 8192 + (size_t)(it * 256 + threadIdx.x) * 16) = r[g * 2 + it];
}

extern "C" int probe_launch(int mode, const void *in, void *out, int nblocks)
{
    const int grid = nblocks * 16;
    switch (mode) {
    case 0: hipLaunchKernelGGL(linear_copy, dim3(grid), dim3(256), 0, 0, (const char *)in, (char *)out); break;
    case 1: hipLaunchKernelGGL(tile_copy<0>, dim3(grid), dim3(256), 0, 0, (const char *)in, (char *)out); break;
    case 2: hipLaunchKernelGGL(tile_copy<1>, dim3(grid), dim3(256), 0, 0, (const char *)in, (char *)out); break;
    case 3: hipLaunchKernelGGL(tilemajor_copy, dim3(grid), dim3(256), 0, 0, (const char *)in, (char *)out); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
