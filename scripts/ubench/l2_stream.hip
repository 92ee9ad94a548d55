// Micro-benchmark: per-CU rate of streaming a weight matrix that is resident in L2 straight into registers (the fused-MLP
// kernels' only memory traffic), by access pattern and loads in flight.  One 512-thread workgroup per CU, every workgroup
// reads the same `bytes` (like the 256 workgroups of an mlp3 launch read the same weights).
//   pattern 0: dwordx2, a 16-lane group reads one 128-byte line, 4 lines (rows 1 KB apart) per instruction  (forward, PAIR)
//   pattern 1: dwordx4, 16 lanes x 16 B = 256 B contiguous, 4 rows 1 KB apart per instruction
//   pattern 2: dwordx4, 64 lanes x 16 B = 1 KB contiguous per instruction
//   pattern 3: dword,   16 lanes x 4 B = 64 B, 4 rows per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int PAT, int INFLIGHT>
__global__ __launch_bounds__(512) void k(const float* __restrict__ w, int rows /* of 256 floats */, int reps, float* out,
                                         unsigned long long* clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        // the matrix is [rows][256]; a wave owns 32 columns (patterns 0, 3) or a share of the rows (1, 2)
        if (PAT == 0) {
            const float* p = w + (size_t)(4 * kq) * 256 + 32 * wave + 2 * n;
            for (int r = 0; r < rows; r += 16 * INFLIGHT / 4) {
                f32x2 v[INFLIGHT];
#pragma unroll
                for (int i = 0; i < INFLIGHT; ++i) v[i] = *reinterpret_cast<const f32x2*>(p + (size_t)(r + 16 * (i / 4) + (i % 4)) * 256);
#pragma unroll
                for (int i = 0; i < INFLIGHT; ++i) { acc[0] += v[i][0]; acc[1] += v[i][1]; }
            }
        } else if (PAT == 3) {
            const float* p = w + (size_t)(4 * kq) * 256 + 32 * wave + n;
            for (int r = 0; r < rows; r += 16 * INFLIGHT / 8) {
                float v[INFLIGHT];
#pragma unroll
                for (int i = 0; i < INFLIGHT; ++i) v[i] = p[(size_t)(r + 16 * (i / 8) + (i % 4)) * 256 + 16 * ((i / 4) % 2)];
#pragma unroll
                for (int i = 0; i < INFLIGHT; ++i) acc[0] += v[i];
            }
        } else if (PAT == 1) {
            // wave owns 64 columns (wave % 4) and half of the rows (wave / 4)
            const float* p = w + (size_t)(kq + (wave / 4) * (rows / 2)) * 256 + 64 * (wave % 4) + 4 * n;
            for (int r = 0; r < rows / 2; r += 4 * INFLIGHT) {
                f32x4 v[INFLIGHT];
#pragma unroll
                for (int i = 0; i < INFLIGHT; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(r + 4 * i) * 256);
#pragma unroll
                for (int i = 0; i < INFLIGHT; ++i) acc += v[i];
            }
        } else {
            // wave owns rows/8 consecutive rows, reads them 1 KB (one row) per instruction
            const float* p = w + (size_t)(wave * (rows / 8)) * 256 + 4 * lane;
            for (int r = 0; r < rows / 8; r += INFLIGHT) {
                f32x4 v[INFLIGHT];
#pragma unroll
                for (int i = 0; i < INFLIGHT; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(r + i) * 256);
#pragma unroll
                for (int i = 0; i < INFLIGHT; ++i) acc += v[i];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[threadIdx.x] = acc[0];
}

template <int PAT, int INFLIGHT>
void run(const char* name, const float* w, int rows, int blocks, float* out, unsigned long long* clk) {
    const int reps = 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<PAT, INFLIGHT>), dim3(blocks), dim3(512), 0, 0, w, rows, reps, out, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    unsigned long long c0; (void)hipMemcpy(&c0, clk, 8, hipMemcpyDeviceToHost);
    const double bytes = (double)rows * 1024.0 * reps;
    printf("%-44s in flight/wave %2d  blocks %3d : %8.1f us  %6.1f KB/us per CU (%5.1f B/clk at 2.1 GHz)  wg0 ticks %llu\n", name, INFLIGHT, blocks,
           best * 1e3f, bytes / 1024.0 / (best * 1e3), bytes / (best * 1e-3 * 2.1e9), c0);
}

int main() {
    const int rows = 768;                     // 768 KB: the three matrices of an mlp3 launch, resident in every XCD's L2
    float *w, *out; unsigned long long* clk;
    (void)hipMalloc(&w, (size_t)rows * 1024); (void)hipMemset(w, 0, (size_t)rows * 1024);
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&clk, 8 * 1024);
    for (int blocks : {1, 16, 256}) {
        run<0, 16>("dwordx2 4 lines/instr (mlp3 forward)", w, rows, blocks, out, clk);
        run<0, 32>("dwordx2 4 lines/instr (mlp3 forward)", w, rows, blocks, out, clk);
        run<0, 48>("dwordx2 4 lines/instr (mlp3 forward)", w, rows, blocks, out, clk);
        run<3, 32>("dword 4 x 64 B/instr", w, rows, blocks, out, clk);
        run<1, 8>("dwordx4 4 x 256 B/instr", w, rows, blocks, out, clk);
        run<1, 16>("dwordx4 4 x 256 B/instr", w, rows, blocks, out, clk);
        run<1, 24>("dwordx4 4 x 256 B/instr", w, rows, blocks, out, clk);
        run<2, 8>("dwordx4 1 KB contiguous/instr", w, rows, blocks, out, clk);
        run<2, 16>("dwordx4 1 KB contiguous/instr", w, rows, blocks, out, clk);
        run<2, 24>("dwordx4 1 KB contiguous/instr", w, rows, blocks, out, clk);
    }
    return 0;
}
