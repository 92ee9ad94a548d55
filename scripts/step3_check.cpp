// A/B harness for the PPO step kernels through the C ABI (no torch): the same random minibatch through
// ts_ppo_grad in step mode 2 (fp32 MFMA) and mode 3 (split-bf16 MFMA); prints the largest gradient difference per
// parameter block relative to the block's largest gradient, the loss sums, and HIP-event timings of both.
//   hipcc -O2 scripts/step3_check.cpp -Iinclude -Ltianshou_amd/lib -ltsengine -Wl,-rpath,$PWD/tianshou_amd/lib -o /tmp/s3c
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "tsengine.h"

#define CK(x) do { int rc_ = (x); if (rc_ != 0) { printf("FAILED %s -> %d: %s\n", #x, rc_, ts_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename T>
T* to_dev(const std::vector<T>& v) {
    T* p = nullptr;
    if (hipMalloc(&p, v.size() * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return p;
}

int main(int argc, char** argv) {
    const int64_t obs = argc > 1 ? atoi(argv[1]) : 17, act = argc > 2 ? atoi(argv[2]) : 6;
    const int64_t n = 1 << 17, rows = argc > 3 ? atoll(argv[3]) : 65536;
    const int adv_norm = argc > 4 ? atoi(argv[4]) : 1;
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::mt19937_64 rng(1234);
    std::normal_distribution<float> N(0.f, 1.f);
    const int64_t P = ts_ppo_param_count(obs, act);
    std::vector<float> params(P);
    // layout (tsengine.h): W1[64][obs] b1 W2[64][64] b2 Wmu[act][64] bmu sigma | critic W1 b1 W2 b2 Wv bv
    for (auto& v : params) v = 0.2f * N(rng);
    std::vector<float> h_obs(n * obs), h_act(n * act), h_adv(n), h_ret(n), h_lp(n), h_v(n);
    for (auto& v : h_obs) v = N(rng);
    for (auto& v : h_act) v = N(rng);
    for (auto& v : h_adv) v = N(rng);
    for (auto& v : h_ret) v = N(rng);
    for (auto& v : h_lp) v = -0.92f * act - 0.5f * std::fabs(N(rng)) * act * 0.3f;
    for (auto& v : h_v) v = 0.3f * N(rng);
    std::vector<int64_t> perm(n);
    for (int64_t i = 0; i < n; ++i) perm[i] = i;
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<float> stats = {0.013f, 0.987f};

    ts_workspace* ws = nullptr;
    CK(ts_workspace_create(&ws, 0, (size_t)1 << 31));
    float *d_params = to_dev(params), *d_obs = to_dev(h_obs), *d_act = to_dev(h_act), *d_adv = to_dev(h_adv),
          *d_ret = to_dev(h_ret), *d_lp = to_dev(h_lp), *d_v = to_dev(h_v), *d_stats = to_dev(stats);
    int64_t* d_perm = to_dev(perm);
    const int64_t rw = ts_ppo_record_width(obs, act);
    float* d_rec; HK(hipMalloc(&d_rec, sizeof(float) * n * rw));
    CK(ts_ppo_pack_batch(d_obs, d_act, d_adv, d_ret, d_lp, d_v, n, obs, act, d_rec, nullptr));
    ts_ppo_hparams hp{};
    hp.eps_clip = 0.2; hp.dual_clip = 3.0; hp.vf_coef = 0.25; hp.ent_coef = 0.01; hp.max_grad_norm = 0.5;
    hp.lr = 3e-4; hp.beta1 = 0.9; hp.beta2 = 0.999; hp.adam_eps = 1e-8; hp.value_clip = 1; hp.adv_norm = adv_norm; hp.algo = 0;

    float *d_g[2], *d_parts[2];
    std::vector<float> g[2], parts[2];
    hipEvent_t e0, e1; HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    const int modes[2] = {2, getenv("S3_MODE") ? atoi(getenv("S3_MODE")) : 3};
    for (int m = 0; m < 2; ++m) {
        HK(hipMalloc(&d_g[m], sizeof(float) * (P + 8))); HK(hipMalloc(&d_parts[m], sizeof(float) * 8));
        HK(hipMemset(d_g[m], 0, sizeof(float) * (P + 8)));
        CK(ts_ppo_set_step_mode(modes[m]));
        CK(ts_ppo_invalidate_image(ws));
        CK(ts_ppo_grad(ws, d_params, obs, act, d_rec, n, d_perm, rows, rows, d_stats, &hp, d_g[m], d_parts[m], nullptr));
        HK(hipDeviceSynchronize());
        g[m].resize(P); parts[m].resize(4);
        HK(hipMemcpy(g[m].data(), d_g[m], sizeof(float) * P, hipMemcpyDeviceToHost));
        HK(hipMemcpy(parts[m].data(), d_parts[m], sizeof(float) * 4, hipMemcpyDeviceToHost));
        // timing
        for (int k = 0; k < 5; ++k) CK(ts_ppo_grad(ws, d_params, obs, act, d_rec, n, d_perm, rows, rows, d_stats, &hp, d_g[m], d_parts[m], nullptr));
        HK(hipDeviceSynchronize());
        const int reps = 50;
        HK(hipEventRecord(e0));
        for (int k = 0; k < reps; ++k) CK(ts_ppo_grad(ws, d_params, obs, act, d_rec, n, d_perm, rows, rows, d_stats, &hp, d_g[m], d_parts[m], nullptr));
        HK(hipEventRecord(e1)); HK(hipEventSynchronize(e1));
        float ms; HK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d: %.2f us per ts_ppo_grad (step kernel + slab reduction), %lld rows\n", modes[m], ms * 1e3f / reps, (long long)rows);
        if (getenv("S3_PHASES")) {
            static const char* names[18] = {"start", "staged", "A:trunk", "A:loss", "A:headgrad", "A:dH1", "A:tiles_A", "A:dW2", "A:dW1/misc",
                                            "C:staged", "C:trunk", "C:loss", "C:headgrad", "C:dH1", "C:tiles_A", "C:dW2", "C:dW1/misc", "end"};
            for (int trial = 0; trial < 3; ++trial) {
                int64_t cyc[32] = {0};
                CK(ts_debug_ppo_step_cycles(ws, d_params, obs, act, d_rec, d_perm + 1000 * trial, rows, &hp, cyc, 32, nullptr));
                if (trial == 0) continue;
                printf("  mode %d trial %d total %lld cycles:", modes[m], trial, (long long)(cyc[17] - cyc[0]));
                for (int k = 1; k < 18; ++k) printf(" %s +%lld", names[k], (long long)(cyc[k] - cyc[k - 1]));
                printf("\n");
            }
        }
    }
    struct Blk { const char* name; int64_t lo, hi; };
    const int64_t H = 64;
    int64_t o = 0;
    std::vector<Blk> blks;
    auto add = [&](const char* nm, int64_t len) { blks.push_back({nm, o, o + len}); o += len; };
    add("a.W1", H * obs); add("a.b1", H); add("a.W2", H * H); add("a.b2", H); add("a.Wmu", act * H); add("a.bmu", act); add("a.sigma", act);
    add("c.W1", H * obs); add("c.b1", H); add("c.W2", H * H); add("c.b2", H); add("c.Wv", H); add("c.bv", 1);
    if (o != P) { printf("layout mismatch %lld vs %lld\n", (long long)o, (long long)P); return 1; }
    int bad = 0;
    for (auto& b : blks) {
        double mx = 0, md = 0; int64_t arg = -1;
        for (int64_t i = b.lo; i < b.hi; ++i) {
            mx = std::fmax(mx, std::fabs((double)g[0][i]));
            const double dd = std::fabs((double)g[0][i] - (double)g[1][i]);
            if (dd > md) { md = dd; arg = i - b.lo; }
        }
        const double rel = md / (mx > 0 ? mx : 1);
        printf("%-8s max|g| %.3e  max|diff| %.3e  rel %.2e  (at %lld)%s\n", b.name, mx, md, rel, (long long)arg, rel > 2e-5 ? "   <-- MISMATCH" : "");
        if (rel > 2e-5 || std::isnan(rel)) ++bad;
    }
    printf("loss parts mode2: %.7e %.7e %.7e %.7e\n", parts[0][0], parts[0][1], parts[0][2], parts[0][3]);
    printf("loss parts mode3: %.7e %.7e %.7e %.7e\n", parts[1][0], parts[1][1], parts[1][2], parts[1][3]);
    printf(bad ? "RESULT: MISMATCH in %d blocks\n" : "RESULT: OK (%d)\n", bad);
    return bad ? 2 : 0;
}
