// ts_ppo_step3.h -- third-generation PPO step kernel: fp32 GEMMs on the bf16 matrix cores through a three-way
// operand split.  Included by ts_ppo.hip inside its anonymous namespace (shares StepArgs, Dims, the record fetch, the
// slab layout and the loss section's semantics with ppo_step2_kernel).
//
// Why: on gfx950 v_mfma_f32_32x32x2_f32 runs on the SIMD's fp32 vector lanes (64 cycles, 64 flop / cycle / SIMD) and
// does not overlap with VALU work of either resident wave (profiles/r02_ubench_mfma_valu.txt); the bf16 matrix core
// runs v_mfma_f32_32x32x16_bf16 in 32 cycles (1024 flop / cycle / SIMD) BESIDE the VALU
// (profiles/r02_ubench_mfma_bf16.txt).  Every fp32 operand x is written as x = x0 + x1 + x2 with xk = bf16_rne of the
// running remainder (exact: 3 x 8 significant bits cover the 24-bit significand; x - x0 and x - x0 - x1 are exact in
// fp32), and a product sum is evaluated as
//     sum_k a b  ~=  sum_k (a0 b0 + a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0)        (six bf16 MFMAs, fp32 accumulate)
// Every bf16 x bf16 product is exact in fp32; the dropped terms (a1 b2, a2 b1, a2 b2) are bounded by 2^-26 |a b| --
// a quarter of the rounding error an fp32 multiply commits.  Eight K = 2 fp32 MFMAs (512 cycles) become six K = 16
// bf16 MFMAs (192 cycles) that leave the VALU free.
//
// Operand order: K is a summation index, so the k-slot (lane half h, element j) of chunk c is DEFINED to carry
//   * hidden feature 32 (c >> 1) + F(8 (c & 1) + j, h) for the 64-wide layers -- the C/D layout of the producing MFMA
//     (register r of tile t holds feature 32 t + F(r, h)), so activations stay in their lanes: registers (r, r + 1)
//     are converted pairwise (v_cvt_pk_bf16_f32) straight into the B operand;
//   * input column 16 c + 8 h + j (obs | 1 | 0-pad) for layer 1.
// The weights (A operands) come from a ready-made LDS image that holds the three pieces of every weight in exactly
// that order (one ds_read_b128 per piece and chunk); the image is built once per update and refreshed in place by
// ppo_adam_kernel.  The backward product dH1^T = W2^T dZ2^T needs W2 by columns: the lanes gather 16-bit elements from
// the same image (chunk / half strides padded so that the 64 lanes hit 64 distinct dwords modulo the bank count).

namespace s3 {

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int CH2 = 1056;                      // bytes per W2 chunk: 64 lanes x 16 B + HP + 16 (stride = 32 mod 128)
constexpr int HP = 16;                         // gap between the two lane halves of a W2 chunk
constexpr int W2P_PIECE = 8 * CH2;             // [t2][c] chunks of one piece
constexpr int W2P_BYTES = 3 * W2P_PIECE;
constexpr int W1P_OFF = W2P_BYTES;
constexpr int W1P_PIECE = 4 * 1024;            // [t][c] chunks of one piece, lane-linear
constexpr int W1P_BYTES = 3 * W1P_PIECE;
constexpr int F32_OFF = W1P_OFF + W1P_BYTES;   // fp32 tail: b2[64] | head image [h][t][r][8] | SMALL[32]
constexpr int B2_F = 0, WH_F = 64, SMALL_F = 64 + 512;
constexpr int F32_FLOATS = 64 + 512 + 32;
constexpr int IMG_BYTES = F32_OFF + 4 * F32_FLOATS;      // per net
static_assert(IMG_BYTES % 16 == 0 && F32_OFF % 16 == 0, "image alignment");
constexpr int LDS_BYTES = IMG_BYTES + STEP_WAVES * 2 * TILE_SIZE * 4;
static_assert(LDS_BYTES >= T2_FLOATS * 4, "the fp32 gradient tiles overlay image + scratch");
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float lo_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f32(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// three bf16 pieces of the 8 k-slots a lane contributes to one K = 16 chunk (4 packed registers per piece)
struct P3 { u32x4 p[3]; };

struct Pk3 { unsigned p0, p1, p2; };

__device__ __forceinline__ Pk3 split_pair(float a, float b) {
    Pk3 o;
    o.p0 = cvt_pk(a, b);
    const float ra = a - lo_f32(o.p0), rb = b - hi_f32(o.p0);       // exact
    o.p1 = cvt_pk(ra, rb);
    const float sa = ra - lo_f32(o.p1), sb = rb - hi_f32(o.p1);     // exact
    o.p2 = cvt_pk(sa, sb);
    return o;
}

// registers 8u .. 8u+7 of a C/D tile -> chunk u of that tile
__device__ __forceinline__ void split_tile(const f32x16& v, P3 (&out)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            const Pk3 s = split_pair(v[8 * u + 2 * q], v[8 * u + 2 * q + 1]);
            out[u].p[0][q] = s.p0; out[u].p[1][q] = s.p1; out[u].p[2][q] = s.p2;
        }
}

__device__ __forceinline__ f32x16 mfma_bf(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// c += A B for one K = 16 chunk, smallest terms first
__device__ __forceinline__ f32x16 mma6(const P3& a, const P3& b, f32x16 c) {
    c = mfma_bf(a.p[2], b.p[0], c);
    c = mfma_bf(a.p[1], b.p[1], c);
    c = mfma_bf(a.p[0], b.p[2], c);
    c = mfma_bf(a.p[1], b.p[0], c);
    c = mfma_bf(a.p[0], b.p[1], c);
    c = mfma_bf(a.p[0], b.p[0], c);
    return c;
}

// ---- image: parameter -> slot.  code = kind << 24 | byte offset inside [2][IMG_BYTES]; kind 1: bf16 piece 0 of a W2
// element (pieces W2P_PIECE apart), 2: bf16 piece 0 of a W1aug element (W1P_PIECE apart), 3: fp32 slot, 0: none (sigma)
__device__ __forceinline__ int hid_slot_bytes(int f, int& chunk) {   // feature f of a 64-wide layer -> chunk, (h, j)
    const int t = f >> 5, fi = f & 31;
    const int h = (fi >> 2) & 1, r = (fi & 3) | ((fi >> 3) << 2);
    chunk = 2 * t + (r >> 3);
    return h * (512 + HP) + (r & 7) * 2;
}

__device__ __forceinline__ int param_code(int p, const Dims& d) {
    const int net = p >= d.p_actor;
    const int base = net * IMG_BYTES;
    const int w1 = net ? d.c_w1 : d.a_w1, b1 = net ? d.c_b1 : d.a_b1, w2 = net ? d.c_w2 : d.a_w2, b2 = net ? d.c_b2 : d.a_b2;
    if (p < b1) {                                   // W1[row][k]
        const int q = p - w1, row = q / d.obs, k = q - row * d.obs;
        const int off = W1P_OFF + ((row >> 5) * 2 + (k >> 4)) * 1024 + ((row & 31) + 32 * ((k >> 3) & 1)) * 16 + (k & 7) * 2;
        return (2 << 24) | (base + off);
    }
    if (p < w2) {                                   // b1[row] rides as column k = obs
        const int row = p - b1, k = d.obs;
        const int off = W1P_OFF + ((row >> 5) * 2 + (k >> 4)) * 1024 + ((row & 31) + 32 * ((k >> 3) & 1)) * 16 + (k & 7) * 2;
        return (2 << 24) | (base + off);
    }
    if (p < b2) {                                   // W2[f2][f1]
        const int q = p - w2, f2 = q >> 6, f1 = q & 63;
        int chunk;
        const int inner = hid_slot_bytes(f1, chunk);
        const int off = ((f2 >> 5) * 4 + chunk) * CH2 + (f2 & 31) * 16 + inner;
        return (1 << 24) | (base + off);
    }
    if (p < b2 + HID) return (3 << 24) | (base + F32_OFF + 4 * (B2_F + (p - b2)));
    const int q = p - (b2 + HID);                   // head W | head b | sigma
    const int n_head = net ? 1 : d.act;
    if (q < n_head * HID) {
        const int a = q / HID, f = q - a * HID;
        const int t = f >> 5, fi = f & 31, h = (fi >> 2) & 1, r = (fi & 3) | ((fi >> 3) << 2);
        return (3 << 24) | (base + F32_OFF + 4 * (WH_F + ((h * 2 + t) * 16 + r) * ACT_PAD + a));
    }
    const int e = q - n_head * HID;
    if (e < n_head) return (3 << 24) | (base + F32_OFF + 4 * (SMALL_F + (net ? 24 : e)));
    return 0;                                       // sigma_param: two derived slots, written explicitly
}

__device__ __forceinline__ void image_put(char* image, int code, float v) {
    const int kind = code >> 24, off = code & 0xffffff;
    if (kind == 3) {
        *reinterpret_cast<float*>(image + off) = v;
    } else if (kind != 0) {
        const int stride = kind == 1 ? W2P_PIECE : W1P_PIECE;
        const unsigned p0 = cvt_pk(v, 0.f);
        const float r1 = v - lo_f32(p0);
        const unsigned p1 = cvt_pk(r1, 0.f);
        const float r2 = r1 - lo_f32(p1);
        const unsigned p2 = cvt_pk(r2, 0.f);
        *reinterpret_cast<u16*>(image + off) = (u16)p0;
        *reinterpret_cast<u16*>(image + off + stride) = (u16)p1;
        *reinterpret_cast<u16*>(image + off + 2 * stride) = (u16)p2;
    }
}

__device__ __forceinline__ void image_put_sigma(char* image, int k, float sigma_param) {   // finish_small's two slots
    const float sigma = expf(sigma_param);
    float* sm = reinterpret_cast<float*>(image + F32_OFF) + SMALL_F;
    sm[8 + k] = 1.f / (2.f * (sigma * sigma));
    sm[16 + k] = logf(sigma);
}

// one workgroup: zero both images, then every parameter writes its slot(s); inv[p] = code
__global__ __launch_bounds__(1024) void ppo_build_image3_kernel(const float* __restrict__ params, Dims d,
                                                                char* __restrict__ image, int* __restrict__ inv) {
    for (int i = threadIdx.x; i < 2 * IMG_BYTES / 4; i += 1024) reinterpret_cast<int*>(image)[i] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < d.p_total; p += 1024) {
        const int code = param_code(p, d);
        if (inv) inv[p] = code;
        const float v = params[p];
        image_put(image, code, v);
        const int k = p - d.a_sig;
        if (k >= 0 && k < d.act) image_put_sigma(image, k, v);
    }
}

// ---- staging: straight 16-byte copy of one net's image
__device__ __forceinline__ void stage_image3(char* lds, const char* __restrict__ img, int tid) {
    constexpr int N4 = IMG_BYTES / 16, PER = (N4 + STEP_THREADS - 1) / STEP_THREADS;
    f32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int q = tid + STEP_THREADS * k;
        v[k] = reinterpret_cast<const f32x4*>(img)[q < N4 ? q : N4 - 1];      // unconditional clamped loads
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int q = tid + STEP_THREADS * k;
        if (q < N4) reinterpret_cast<f32x4*>(lds)[q] = v[k];
    }
}

// B operand of layer 1: input column 16 c + 8 h + j of the lane's record (obs | 1 | 0)
template <int NC1>
__device__ __forceinline__ void x_pieces(const float* rec_row, int obs, int h, P3 (&xp)[NC1]) {
#pragma unroll
    for (int c = 0; c < NC1; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 16 * c + 8 * h + 2 * q + e;
                const float x = rec_row[k < obs ? k : 0];
                v[e] = k < obs ? x : (k == obs ? 1.f : 0.f);
            }
            const Pk3 s = split_pair(v[0], v[1]);
            xp[c].p[0][q] = s.p0; xp[c].p[1][q] = s.p1; xp[c].p[2][q] = s.p2;
        }
}

// forward trunk: x -> h1 (+ its pieces) -> h2
template <int NC1>
__device__ __forceinline__ void trunk_forward3(const char* L, const P3 (&xp)[NC1], int lane, f32x16 (&h1)[2],
                                               P3 (&h1p)[4], f32x16 (&h2)[2]) {
    const int h = lane >> 5;
    const char* w1 = L + W1P_OFF + lane * 16;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC1; ++c) {
            P3 a;
#pragma unroll
            for (int p = 0; p < 3; ++p) a.p[p] = *reinterpret_cast<const u32x4*>(w1 + p * W1P_PIECE + (t * 2 + c) * 1024);
            acc = mma6(a, xp[c], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fast_tanh(acc[r]);
        h1[t] = acc;
        P3 two[2];
        split_tile(acc, two);
        h1p[2 * t] = two[0];
        h1p[2 * t + 1] = two[1];
    }
    const char* w2 = L + lane * 16 + h * HP;
    const float* b2 = reinterpret_cast<const float*>(L + F32_OFF) + B2_F;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(b2 + 32 * t2 + 8 * g + 4 * h);
            acc[4 * g + 0] = b[0]; acc[4 * g + 1] = b[1]; acc[4 * g + 2] = b[2]; acc[4 * g + 3] = b[3];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            P3 a;
#pragma unroll
            for (int p = 0; p < 3; ++p) a.p[p] = *reinterpret_cast<const u32x4*>(w2 + p * W2P_PIECE + (t2 * 4 + c) * CH2);
            acc = mma6(a, h1p[c], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fast_tanh(acc[r]);
        h2[t2] = acc;
    }
}

// head on the VALU (same arithmetic as head_forward), weights from the image's fp32 tail
template <int NA>
__device__ __forceinline__ void head_forward3(const float* whb, int h, const f32x16 (&h2)[2], float (&out)[NA]) {
    const float* wh = whb + h * (2 * 16 * ACT_PAD);
#pragma unroll
    for (int a = 0; a < NA; ++a) out[a] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* p = wh + (t * 16 + r) * ACT_PAD;
            if constexpr (NA == 1) {
                out[0] += h2[t][r] * p[0];
            } else {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(p);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    out[a] += h2[t][r] * w0[a];
                    if (a + 4 < NA) out[a + 4] += h2[t][r] * w1[a];
                }
            }
            if (NA > 1 && (r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) out[a] += __shfl_xor(out[a], 32, 64);
}

// A operand of dH1^T = W2^T dZ2^T for chunk C of output tile T1: the lane's column of W2 gathered from the forward
// image with 16-bit LDS loads.  Written as ONE asm block per
// chunk -- 24 loads in flight, one wait -- because hipcc serialises the C++ form into 12 load-load-wait-combine round
// trips per chunk (10 k cycles per net).  Rows F(8u + j, 0), j = 0..7, are 16 u + {0, 1, 2, 3, 8, 9, 10, 11}.
template <int T1, int C>
__device__ __forceinline__ void gather_w2_column(unsigned lane_addr, P3& a) {
    constexpr int t = C >> 1, u = C & 1;
    constexpr int B0 = ((0 * 2 + t) * 4 + 2 * T1) * CH2 + 256 * u;      // piece 0; pieces are W2P_PIECE apart
    static_assert(B0 + 2 * W2P_PIECE + 176 < 65536, "DS offset field");
    unsigned l0, l1, l2, l3, l4, l5, l6, l7, l8, l9, l10, l11, h0, h1, h2, h3, h4, h5, h6, h7, h8, h9, h10, h11;
    asm volatile(
        "ds_read_u16 %0, %24 offset:%25\n\t"
        "ds_read_u16 %1, %24 offset:%25+32\n\t"
        "ds_read_u16 %2, %24 offset:%25+128\n\t"
        "ds_read_u16 %3, %24 offset:%25+160\n\t"
        "ds_read_u16 %4, %24 offset:%26\n\t"
        "ds_read_u16 %5, %24 offset:%26+32\n\t"
        "ds_read_u16 %6, %24 offset:%26+128\n\t"
        "ds_read_u16 %7, %24 offset:%26+160\n\t"
        "ds_read_u16 %8, %24 offset:%27\n\t"
        "ds_read_u16 %9, %24 offset:%27+32\n\t"
        "ds_read_u16 %10, %24 offset:%27+128\n\t"
        "ds_read_u16 %11, %24 offset:%27+160\n\t"
        "ds_read_u16 %12, %24 offset:%25+16\n\t"
        "ds_read_u16 %13, %24 offset:%25+48\n\t"
        "ds_read_u16 %14, %24 offset:%25+144\n\t"
        "ds_read_u16 %15, %24 offset:%25+176\n\t"
        "ds_read_u16 %16, %24 offset:%26+16\n\t"
        "ds_read_u16 %17, %24 offset:%26+48\n\t"
        "ds_read_u16 %18, %24 offset:%26+144\n\t"
        "ds_read_u16 %19, %24 offset:%26+176\n\t"
        "ds_read_u16 %20, %24 offset:%27+16\n\t"
        "ds_read_u16 %21, %24 offset:%27+48\n\t"
        "ds_read_u16 %22, %24 offset:%27+144\n\t"
        "ds_read_u16 %23, %24 offset:%27+176\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "=&v"(l4), "=&v"(l5), "=&v"(l6), "=&v"(l7), "=&v"(l8),
          "=&v"(l9), "=&v"(l10), "=&v"(l11), "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(h4), "=&v"(h5),
          "=&v"(h6), "=&v"(h7), "=&v"(h8), "=&v"(h9), "=&v"(h10), "=&v"(h11)
        : "v"(lane_addr), "n"(B0), "n"(B0 + W2P_PIECE), "n"(B0 + 2 * W2P_PIECE)
        : "memory");
    // (d16 / d16_hi loads would fill the packed registers directly, but with SRAM ECC on -- as on this part -- they
    // clear the other half instead of preserving it)
    a.p[0][0] = l0 | (h0 << 16); a.p[0][1] = l1 | (h1 << 16); a.p[0][2] = l2 | (h2 << 16); a.p[0][3] = l3 | (h3 << 16);
    a.p[1][0] = l4 | (h4 << 16); a.p[1][1] = l5 | (h5 << 16); a.p[1][2] = l6 | (h6 << 16); a.p[1][3] = l7 | (h7 << 16);
    a.p[2][0] = l8 | (h8 << 16); a.p[2][1] = l9 | (h9 << 16); a.p[2][2] = l10 | (h10 << 16); a.p[2][3] = l11 | (h11 << 16);
}

template <int T1>
__device__ __forceinline__ f32x16 dh1_tile(unsigned lane_addr, const P3 (&dz2p)[4]) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    P3 a;
    gather_w2_column<T1, 0>(lane_addr, a); acc = mma6(a, dz2p[0], acc);
    gather_w2_column<T1, 1>(lane_addr, a); acc = mma6(a, dz2p[1], acc);
    gather_w2_column<T1, 2>(lane_addr, a); acc = mma6(a, dz2p[2], acc);
    gather_w2_column<T1, 3>(lane_addr, a); acc = mma6(a, dz2p[3], acc);
    return acc;
}

// dH1^T = W2^T dZ2^T, then dZ1 = dH1 (1 - h1^2).  lds_base = LDS byte address of the image.
__device__ __forceinline__ void dh1_backward3(unsigned lds_base, const P3 (&dz2p)[4], const f32x16 (&h1)[2], int lane,
                                              f32x16 (&dz1)[2]) {
    const int i = lane & 31, h = lane >> 5;
    // the lane's own column f1 = 32 t1 + i sits in chunk 2 t1 + (i >> 4), half (i >> 2) & 1, element (i & 3) | (i >> 3 & 1) << 2;
    // the rows it needs are F(r, h) = F(r, 0) + 4 h
    const int j2 = (i & 3) | (((i >> 3) & 1) << 2);
    const unsigned lane_addr = lds_base + (i >> 4) * CH2 + ((i >> 2) & 1) * (512 + HP) + j2 * 2 + h * 64;
    f32x16 acc0 = dh1_tile<0>(lane_addr, dz2p);
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float hv = h1[0][r]; acc0[r] = acc0[r] * (1.f - hv * hv); }
    dz1[0] = acc0;
    f32x16 acc1 = dh1_tile<1>(lane_addr, dz2p);
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float hv = h1[1][r]; acc1[r] = acc1[r] * (1.f - hv * hv); }
    dz1[1] = acc1;
}

// forward, loss and backward of one net for the wave's 32 samples (see net_fwd_bwd: identical loss section)
template <int KS1, int NC1, bool ACTOR>
__device__ __forceinline__ void net_fwd_bwd3(const char* L, float* scratch, const StepArgs& g, const Dims& d,
                                             const TileIn<KS1>& in, const P3 (&xp)[NC1], int lane_in, f32x16 (&h1)[2],
                                             f32x16 (&h2)[2], P3 (&dz2p)[4], f32x16 (&dz1)[2],
                                             float (&gw)[ACTOR ? ACT_PAD : 1], float& misc) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    [[maybe_unused]] constexpr int MK = ACTOR ? 2 : 10;
    constexpr int NA = ACTOR ? ACT_PAD : 1;
    const int i = lane & 31, h = lane >> 5;
    const float* f32t = reinterpret_cast<const float*>(L + F32_OFF);
    P3 h1p[4];
    trunk_forward3<NC1>(L, xp, lane, h1, h1p, h2);
    TS_MARK(g, MK + 0);

    float dout[NA];
    const float w = in.w;
    const float* sm = f32t + SMALL_F;
    float* Qt = scratch;
    if constexpr (ACTOR) {
        float mu[ACT_PAD];
        head_forward3<ACT_PAD>(f32t + WH_F, h, h2, mu);
        const f32x4 b0 = ld4(sm), b1 = ld4(sm + 4), v0 = ld4(sm + 8), v1 = ld4(sm + 12), s0 = ld4(sm + 16), s1 = ld4(sm + 20);
        float dlt[ACT_PAD], inv_var[ACT_PAD];
        float logp = 0.f;
        int n_act = d.act;
        asm volatile("" : "+s"(n_act));
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) {
            const float bm = k < 4 ? b0[k & 3] : b1[k & 3], iv = k < 4 ? v0[k & 3] : v1[k & 3], ls = k < 4 ? s0[k & 3] : s1[k & 3];
            const float m = mu[k] + bm;
            dlt[k] = in.act[k] - m;
            inv_var[k] = 2.f * iv;
            logp += -(dlt[k] * dlt[k]) * iv - ls - (k < n_act ? LOG_SQRT_2PI : 0.f);
        }
        float mean = 0.f, den = 1.f;
        if (g.adv_norm) { mean = g.adv_stats[0]; den = g.adv_stats[1] + 1e-8f; }
        const float A = (in.adv - mean) / den;
        const bool a2c = g.a2c != 0;
        const float ratio = a2c ? 1.f : expf(logp - in.logp_old);
        const float surr1 = ratio * A;
        const float lo = 1.f - g.eps_clip, hi = 1.f + g.eps_clip;
        const float surr2 = fminf(fmaxf(ratio, lo), hi) * A;
        const float clip1 = fminf(surr1, surr2);
        float basek = (surr1 <= surr2) ? A : 0.f;
        const float dA = g.dual_clip * A;
        const bool dual = (g.dual_clip > 0.f) && (A < 0.f);
        float term = dual ? -fmaxf(clip1, dA) : -clip1;
        basek = (dual && !(clip1 >= dA)) ? 0.f : basek;
        term = a2c ? -logp * A : term;
        basek = a2c ? A : basek;
        const float dlogp = -basek * ratio * w;
        const float ent_w = g.ent_coef * w;
        float dsig[ACT_PAD];
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) {
            dout[k] = dlogp * dlt[k] * inv_var[k];
            const float ds = dlogp * (dlt[k] * dlt[k] * inv_var[k] - 1.f) - ent_w;
            dsig[k] = k < n_act ? ds : 0.f;
        }
#pragma unroll
        for (int k = 0; k < ACT_PAD; ++k) Qt[(8 * h + k) * TILE_PITCH + i] = h ? dsig[k] : dout[k];
        if (h == 0) Qt[16 * TILE_PITCH + i] = term * w;
    } else {
        float v[1];
        head_forward3<1>(f32t + WH_F, h, h2, v);
        const float value = v[0] + sm[24];
        const float ret = in.ret;
        const float vf1 = (ret - value) * (ret - value);
        const float vo = in.v_old;
        const float dvo = value - vo;
        const float vclip = vo + fminf(fmaxf(dvo, -g.eps_clip), g.eps_clip);
        const float vf2 = (ret - vclip) * (ret - vclip);
        const float g1 = -2.f * (ret - value);
        const float g2 = (dvo >= -g.eps_clip && dvo <= g.eps_clip) ? -2.f * (ret - vclip) : 0.f;
        const float dv_clip = (vf1 > vf2) ? g1 : ((vf2 > vf1) ? g2 : 0.5f * (g1 + g2));
        const bool vc = g.value_clip != 0;
        const float term = vc ? fmaxf(vf1, vf2) : vf1;
        const float dv = vc ? dv_clip : g1;
        dout[0] = dv * g.vf_coef * w;
        Qt[(16 * h) * TILE_PITCH + i] = h ? term * w : dout[0];
    }
    wave_lds_sync();
    {
        const int row = lane < 16 ? lane : 16;
        const float* rp = Qt + row * TILE_PITCH;
        f32x4 q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = ld4(rp + 4 * k);
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc += (q[k][0] + q[k][1]) + (q[k][2] + q[k][3]);
        const bool mine = ACTOR ? (lane <= 16) : (lane == 0 || lane == 16);
        misc = mine ? sacc : 0.f;
    }
    wave_lds_sync();
    __builtin_amdgcn_sched_barrier(0);
    TS_MARK(g, MK + 1);

    // ---- head weight gradient (unchanged: VALU over the wave's two transposed h2 tiles)
    float* SA = scratch;
    float* SB = scratch + TILE_SIZE;
    tile_write(SA, h2[0], i, h);
    tile_write(SB, h2[1], i, h);
    if (lane < 32) {
        if constexpr (ACTOR) {
            const f32x4 d0 = {dout[0], dout[1], dout[2], dout[3]};
            const f32x4 d1 = {dout[4], dout[5], dout[6], dout[7]};
            *reinterpret_cast<f32x4*>(SA + lane * TILE_PITCH + 32) = d0;
            *reinterpret_cast<f32x4*>(SB + lane * TILE_PITCH + 32) = d1;
        } else {
            SA[lane * TILE_PITCH + 32] = dout[0];
        }
    }
    wave_lds_sync();
    {
        const float* rowp = (lane < 32 ? SA : SB) + (lane & 31) * TILE_PITCH;
#pragma unroll
        for (int a = 0; a < NA; ++a) gw[a] = 0.f;
#pragma unroll 2
        for (int q = 0; q < 8; ++q) {
            const f32x4 hv = ld4(rowp + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int smp = 4 * q + e;
                if constexpr (ACTOR) {
                    const f32x4 d0 = ld4(SA + smp * TILE_PITCH + 32);
                    const f32x4 d1 = ld4(SB + smp * TILE_PITCH + 32);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        gw[a] += d0[a] * hv[e];
                        gw[a + 4] += d1[a] * hv[e];
                    }
                } else {
                    gw[0] += SA[smp * TILE_PITCH + 32] * hv[e];
                }
            }
        }
    }
    wave_lds_sync();
    __builtin_amdgcn_sched_barrier(0);
    TS_MARK(g, MK + 2);

    // ---- dZ2 = (dout . Whead) * (1 - h2^2)   (in place in h2)
    {
        const float* wh = f32t + WH_F + h * (2 * 16 * ACT_PAD);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* p = wh + (t * 16 + r) * ACT_PAD;
                float dh;
                if constexpr (ACTOR) {
                    const f32x4 w0 = ld4(p);
                    const f32x4 w1 = ld4(p + 4);
                    dh = dout[0] * w0[0] + dout[1] * w0[1] + dout[2] * w0[2] + dout[3] * w0[3] +
                         dout[4] * w1[0] + dout[5] * w1[1] + dout[6] * w1[2] + dout[7] * w1[3];
                } else {
                    dh = dout[0] * p[0];
                }
                const float hv = h2[t][r];
                h2[t][r] = dh * (1.f - hv * hv);
                if (ACTOR && (r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            P3 two[2];
            split_tile(h2[t], two);
            dz2p[2 * t] = two[0];
            dz2p[2 * t + 1] = two[1];
        }
        dh1_backward3((unsigned)(size_t)L, dz2p, h1, lane, dz1);
    }
    __builtin_amdgcn_sched_barrier(0);
    TS_MARK(g, MK + 3);
}


// (Stage 2 -- the weight gradients on the bf16 cores as well -- was built, measured slower than the fp32 tiles of net_wgrad
// (68.0 us per ts_ppo_grad against 62.9, profiles/r02_step3_stage2_phases.txt: 336 two-byte scatter stores per wave and
// net, eight barriers instead of four) and removed in round 3; `git show 9a55f15:tianshou_amd/csrc/ts_ppo_step3.h` has it.)

template <int KS1>
__global__ __launch_bounds__(STEP_THREADS, 2) void ppo_step3_kernel(StepArgs g, Dims d) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NC1 = (2 * KS1 + 15) / 16;
    char* L = reinterpret_cast<char*>(lds);
    const char* image = reinterpret_cast<const char*>(g.image);
    const int lane0 = threadIdx.x & 63, wave0 = threadIdx.x >> 6;
    const Slab2 SL = slab2_layout(d.act, 2 * KS1);

    const int64_t n_tiles = (g.n_rows + 31) / 32;
    const int64_t per_iter = (int64_t)gridDim.x * STEP_WAVES;
    const int64_t n_iter = (n_tiles + per_iter - 1) / per_iter;
    const int64_t tile0 = (int64_t)blockIdx.x * STEP_WAVES + wave0;
    float* slab = g.slabs + (int64_t)blockIdx.x * g.slab_w;

    TS_MARK(g, 0);
    for (int64_t it = 0; it < n_iter; ++it) {
        int lane = lane0, wave = wave0;
        asm volatile("" : "+v"(lane), "+v"(wave));
        float* scratch = reinterpret_cast<float*>(L + IMG_BYTES) + wave * (2 * TILE_SIZE);
        const RowId row0 = row_fetch(g, it * per_iter + tile0, lane);
        const RecFetch<KS1> f = rec_fetch<KS1>(g, row0, lane);
        if (it > 0) __syncthreads();
        stage_image3(L, image, 64 * wave + lane);
        P3 xp[NC1];
        TileIn<KS1> in;
        {
            // rec_commit with the layer-1 operand pieces taken from the parked record
            constexpr int REC_FETCH = RecFetch<KS1>::N;
            const int i = lane & 31, h = lane >> 5;
            const int parts = g.rec_w >> 2;
            const int total = 32 * parts;
#pragma unroll
            for (int k = 0; k < REC_FETCH; ++k) {
                const int q = lane + 64 * k;
                if (q < total) *reinterpret_cast<f32x4*>(scratch + q * 4) = f.v[k];
            }
            wave_lds_sync();
            const float* r = scratch + i * g.rec_w;
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const int k = KS1 * h + s;
                float v = 0.f;
                if (k < d.obs) v = r[k];
                else if (k == d.obs) v = 1.f;
                in.x[s] = v;
            }
            x_pieces<NC1>(r, d.obs, h, xp);
#pragma unroll
            for (int k = 0; k < ACT_PAD; ++k) in.act[k] = (k < d.act) ? r[d.obs + k] : 0.f;
            const float* aux = r + d.obs + d.act;
            in.adv = aux[0];
            in.ret = aux[1];
            in.logp_old = aux[2];
            in.v_old = aux[3];
            in.w = f.w;
            wave_lds_sync();
        }
        __syncthreads();
        TS_MARK(g, 1);
        const bool first = it == 0;
        f32x16 h1[2], h2[2], dz1[2];
        P3 dz2p[4];
        float misc;
        {
            float gw[ACT_PAD];
            net_fwd_bwd3<KS1, NC1, true>(L, scratch, g, d, in, xp, lane, h1, h2, dz2p, dz1, gw, misc);
            net_wgrad<KS1, true>(lds, g, d, in, h1, h2, dz1, gw, misc, wave, lane, slab, SL, first);
        }
        __syncthreads();
        TS_MARK(g, 18);
        {
            int tid = 64 * wave + lane;
            asm volatile("" : "+v"(tid));
            stage_image3(L, image + IMG_BYTES, tid);
        }
        __syncthreads();
        TS_MARK(g, 9);
        {
            float gw[1];
            net_fwd_bwd3<KS1, NC1, false>(L, scratch, g, d, in, xp, lane, h1, h2, dz2p, dz1, gw, misc);
            net_wgrad<KS1, false>(lds, g, d, in, h1, h2, dz1, gw, misc, wave, lane, slab, SL, first);
        }
    }
    TS_MARK(g, 17);
}

}  // namespace s3
