// How the SHAPE of a wave's 1 KB store instruction prices a streaming write on MI355X: the matrix-core C layout gives every
// buffer_store_dwordx4 sixteen 64-byte row segments (lane (m, g) -> row m, 16 bytes at column 4 g of a 16-column tile; rows 512 bytes
// apart) -- what every strip / attention kernel of this repo stores --, against 8 x 128 B, 4 x 256 B and 1 x 1 KB per instruction.
// Same bytes, same number of instructions, 256 MB per launch.    hipcc --offload-arch=gfx950 -O3 store_shape_probe.hip -o store_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// a [rows][128] fp32 matrix; a workgroup of 4 waves owns 64 rows, a wave 16 rows (as the strip kernels): SEG = bytes per row segment of one
// store instruction (64: one 16-column tile per instruction, 8 instructions per 16 rows; 128: two tiles x 8 rows; 512: a whole row x 2 rows)
template <int SEG, bool LOAD>
__global__ __launch_bounds__(256) void probe(float* __restrict__ out, const float* __restrict__ in, int rows) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long row0 = ((long long)blockIdx.x * 4 + w) * 16;
    if (row0 >= rows) return;
    constexpr int LPS = SEG / 16;                 // lanes per segment
    constexpr int RPI = 64 / LPS;                 // rows per instruction
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // instruction i covers rows [r0, r0 + RPI) x columns [c0, c0 + SEG / 4) of the wave's 16 x 128 block
        const int per_row = 512 / SEG;            // instructions per group of RPI rows
        const int r0 = (i / per_row) * RPI, c0 = (i % per_row) * (SEG / 4);
        const long long off = (row0 + r0 + lane / LPS) * 128 + c0 + (lane % LPS) * 4;
        if (LOAD) v[i] = *(const f32x4*)(in + off);
        else v[i] = f32x4{(float)lane, (float)i, 1.f, 2.f};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int per_row = 512 / SEG;
        const int r0 = (i / per_row) * RPI, c0 = (i % per_row) * (SEG / 4);
        const long long off = (row0 + r0 + lane / LPS) * 128 + c0 + (lane % LPS) * 4;
        if (LOAD) v[i][0] += 1.f;
        *(f32x4*)(out + off) = v[i];
    }
}
template <int SEG, bool LOAD> static void run(float* out, float* in, int rows, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = rows / 64;
    for (int i = 0; i < 3; ++i) probe<SEG, LOAD><<<grid, 256>>>(out, in, rows);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) probe<SEG, LOAD><<<grid, 256>>>(out, in, rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)rows * 512 * (LOAD ? 2 : 1);
    printf("%-28s segments of %4d B: %7.1f us per launch, %5.2f TB/s (%s)\n", what, SEG, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12,
           LOAD ? "read + write" : "write");
}
int main() {
    for (int rows : {25600, 51200, 524288}) {      // 13 MB (the live rows of cfg 2: one tensor), 26 MB, 268 MB
        float *out, *in;
        hipMalloc(&out, (size_t)rows * 512); hipMalloc(&in, (size_t)rows * 512);
        hipMemset(in, 0, (size_t)rows * 512);
        printf("rows %d (%.1f MB)\n", rows, rows * 512 / 1e6);
        run<64, false>(out, in, rows, "store only");
        run<128, false>(out, in, rows, "store only");
        run<256, false>(out, in, rows, "store only");
        run<512, false>(out, in, rows, "store only");
        run<64, true>(out, in, rows, "copy");
        run<128, true>(out, in, rows, "copy");
        run<512, true>(out, in, rows, "copy");
        hipFree(out); hipFree(in);
    }
    return 0;
}
