// K5 / K6: the device side of `microphaser filter` (reference: src/peptides.rs:188-709).
//   K5 translates the mutant / normal nucleotide windows of the neopeptide table and tests every peptide of the tumor
//      protein against the reference peptidome (sorted 5-bit keys, binary search) - byte / integer work.
//   K6 evaluates, per group of records that share a variant region, the binomial likelihood grid, its Simpson integral
//      in log space and the iterative 95 % credible-interval search - f64, transcendental-bound, one thread per group.
//      Restates statrs 0.15 Binomial::pmf / ln_binomial / ln_gamma and bio 0.34 LogProb::{ln_simpsons_integrate_exp,
//      ln_sum_exp}; ln(k!) for k <= 170 comes from a host-built table (ln of the cached f64 factorials, as statrs does).
#include <hip/hip_runtime.h>

#include "kernels_filter.hpp"

namespace mp {

[[noreturn]] void throw_hip(hipError_t e, const char* file, int line);
#define HIP_OK_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__); } while (0)

__constant__ char CODON_AA_F[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVVXYXYSSSSXCWCLFLF";

__device__ __forceinline__ int base2f(uint8_t c, bool complement) {
    if (c >= 'a' && c <= 'z') c -= 32;
    int b;
    switch (c) { case 'A': b = 0; break; case 'C': b = 1; break; case 'G': b = 2; break; case 'T': b = 3; break; default: return -1; }
    return complement ? 3 - b : b;
}

__global__ __launch_bounds__(256) void k5_translate_records(const uint8_t* __restrict__ nt, const uint64_t* __restrict__ nt_off,
                                                            const uint32_t* __restrict__ nt_len, const uint8_t* __restrict__ rev,
                                                            const uint64_t* __restrict__ aa_off, uint64_t n_seq, uint32_t L,
                                                            const uint64_t* __restrict__ ref_keys, uint64_t n_ref, uint8_t* __restrict__ aa,
                                                            uint8_t* __restrict__ flags, uint32_t* __restrict__ err) {
    const uint64_t s = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (s >= n_seq) return;
    const uint32_t len = nt_len[s];
    if (len == 0) return;                      // empty normal_sequence: no protein (:296-299)
    if (len < 2) { atomicOr(err, 2u); return; }  // `r.len() - 2` underflows (:139)
    const uint8_t* q = nt + nt_off[s];
    const bool rc = rev[s] != 0;
    uint8_t* out = aa + aa_off[s];
    const uint32_t ncod = len > 2 ? (len - 2 + 2) / 3 : 0;   // i = 0, 3, ... while i < len - 2
    const uint64_t kmask = L >= 12 ? ~0ull >> 4 : ((1ull << (5 * L)) - 1ull);
    uint64_t key = 0;
    uint32_t x_dist = 0xFFFFFFFFu;   // codons since the last 'X'
    bool bad = false;
    for (uint32_t j = 0; j < ncod; j++) {
        int b0, b1, b2;
        if (!rc) { b0 = base2f(q[3 * j], false); b1 = base2f(q[3 * j + 1], false); b2 = base2f(q[3 * j + 2], false); }
        else { b0 = base2f(q[len - 1 - 3 * j], true); b1 = base2f(q[len - 2 - 3 * j], true); b2 = base2f(q[len - 3 - 3 * j], true); }
        char a = '?';
        if ((b0 | b1 | b2) < 0) bad = true; else a = CODON_AA_F[16 * b0 + 4 * b1 + b2];
        out[j] = uint8_t(a);
        key = ((key << 5) | uint64_t((a - 'A') & 31)) & kmask;
        x_dist = a == 'X' ? 0u : (x_dist == 0xFFFFFFFFu ? x_dist : x_dist + 1);
        if (j + 1 >= L) {  // peptide i = j + 1 - L is complete
            uint8_t f = (x_dist != 0xFFFFFFFFu && x_dist < L) ? 1 : 0;
            uint64_t lo = 0, hi = n_ref;
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (ref_keys[mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < n_ref && ref_keys[lo] == key) f |= 2;
            flags[aa_off[s] + (j + 1 - L)] = f;
        }
    }
    if (bad) atomicOr(err, 1u);
}

void device_translate_records(const uint8_t* d_nt, const uint64_t* d_nt_off, const uint32_t* d_nt_len, const uint8_t* d_rev,
                              const uint64_t* d_aa_off, uint64_t n_seq, uint32_t L, const uint64_t* d_ref_keys, uint64_t n_ref,
                              uint8_t* d_aa, uint8_t* d_flags, uint32_t* d_err, hipStream_t stream) {
    if (!n_seq) return;
    dim3 grid(uint32_t((n_seq + 255) / 256)), block(256);
    hipLaunchKernelGGL(k5_translate_records, grid, block, 0, stream, d_nt, d_nt_off, d_nt_len, d_rev, d_aa_off, n_seq, L, d_ref_keys, n_ref,
                       d_aa, d_flags, d_err);
    HIP_OK_(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------- K6
struct GroupView {
    const double* alt;
    const uint32_t* depth;
    uint32_t n;
    const double* ln_fact;
};

__device__ double ln_gamma_dev(double x) {  // statrs gamma::ln_gamma, x >= 0.5 branch (x = n + 1 > 171 here)
    const double R = 10.900511;
    const double DK[11] = {2.48574089138753565546e-5, 1.05142378581721974210,   -3.45687097222016235469, 4.51227709466894823700,
                           -2.98285225323576655721,   1.05639711577126713077,   -1.95428773191645869583e-1, 1.70970543404441224307e-2,
                           -5.71926117404305781283e-4, 4.63399473359905636708e-6, -2.71994908488607703910e-9};
    const double LN_2_SQRT_E_OVER_PI = 0.6207822376352452223455184457816472122518527279025978;
    double s = DK[0];
    for (int i = 1; i < 11; i++) s += DK[i] / (x + double(i) - 1.0);
    return log(s) + LN_2_SQRT_E_OVER_PI + (x - 0.5) * log((x - 0.5 + R) / 2.718281828459045235360287471352662497757);
}
__device__ __forceinline__ double ln_factorial_dev(const double* ln_fact, uint64_t x) {
    return x <= 170 ? ln_fact[x] : ln_gamma_dev(double(x) + 1.0);
}
__device__ double binomial_pmf_dev(const double* ln_fact, double p, uint64_t n, uint64_t x) {
    if (x > n) return 0.0;
    if (p == 0.0) return x == 0 ? 1.0 : 0.0;
    if (p == 1.0) return x == n ? 1.0 : 0.0;
    const double lb = ln_factorial_dev(ln_fact, n) - ln_factorial_dev(ln_fact, x) - ln_factorial_dev(ln_fact, n - x);
    return exp(lb + double(x) * log(p) + double(n - x) * log(1.0 - p));
}
__device__ __forceinline__ uint64_t round_to_u64_dev(double v) {
    const double r = round(v);
    if (!(r > 0.0)) return 0;
    if (r >= 18446744073709551615.0) return ~0ull;
    return uint64_t(r);
}
__device__ double density_dev(const GroupView& g, double theta) {  // density (:188-201)
    double prob = 1.0;
    for (uint32_t i = 0; i < g.n; i++) prob *= binomial_pmf_dev(g.ln_fact, theta, g.depth[i], round_to_u64_dev(g.alt[i]));
    return prob;
}
// LogProb::ln_simpsons_integrate_exp(|_, v| ln(density(v)) - shift, a, b, n) incl. ln_sum_exp; n <= 99
__device__ double ln_simpson_dev(const GroupView& g, double shift, double a, double b, int n) {
    const double NEG_INF = -__builtin_huge_val();
    const double step = (b - a) / double(n - 1);
    // pass 1: the maximum term and its index (first maximum wins); order of `probs`: interior points 1..n-2, then a, then b
    double pmax = NEG_INF;
    int imax = 0;
    auto term = [&](int k) -> double {  // k = position in `probs`
        if (k < n - 2) {
            const int i = k + 1;
            return log(density_dev(g, a + step * double(i))) - shift + log(double(2 + (i % 2) * 2));
        }
        return log(density_dev(g, k == n - 2 ? a : b)) - shift;
    };
    for (int k = 0; k < n; k++) {
        const double t = term(k);
        if (k == 0 || t > pmax) { pmax = t; imax = k; }
    }
    double lse;
    if (pmax == NEG_INF) lse = NEG_INF;
    else if (pmax == __builtin_huge_val()) lse = pmax;
    else {
        double s = 0.0;
        for (int k = 0; k < n; k++) {
            if (k == imax) continue;
            const double t = term(k);
            if (t != NEG_INF) s += exp(t - pmax);
        }
        lse = pmax + log1p(s);
    }
    return lse + log(b - a) - log(double(n - 1)) - log(3.0);
}

__global__ __launch_bounds__(64) void k6_credible_intervals(const uint64_t* __restrict__ grp_off, const uint8_t* __restrict__ grp_final,
                                                            const double* __restrict__ alt, const uint32_t* __restrict__ depth, uint64_t n_groups,
                                                            const double* __restrict__ ln_fact, CredibleInterval* __restrict__ out) {
    const uint64_t gi = uint64_t(blockIdx.x) * 64 + threadIdx.x;
    if (gi >= n_groups) return;
    GroupView g;
    g.alt = alt + grp_off[gi];
    g.depth = depth + grp_off[gi];
    g.n = uint32_t(grp_off[gi + 1] - grp_off[gi]);
    g.ln_fact = ln_fact;
    CredibleInterval ci;
    ci.status = 0;
    // prob_func (:203-219) + max_by(partial_cmp): the last maximum wins
    uint32_t ml = 0;
    double best = 0.0;
    for (uint32_t t = 0; t < 101; t++) {
        const double p = density_dev(g, double(t) * 0.01);
        if (p != p) ci.status = 1;
        if (t == 0 || p >= best) { best = p; ml = t; }
    }
    const double r = ln_simpson_dev(g, 0.0, 0.0, 1.0, 99);
    const double L95 = log(0.95), L96 = log(0.96);
    double a = ml < 10 ? 0.0 : double(ml - 10) * 0.01;
    double b = ml > 90 ? 1.0 : double(ml + 10) * 0.01;
    double p = -__builtin_huge_val();
    if (!grp_final[gi]) {
        double a_old = double(ml) * 0.01, b_old = double(ml) * 0.01;
        for (int counter = 0; counter != 50; counter++) {
            if (p < L95) {
                a_old = a;
                a = a < 0.1 ? 0.0 : a - 0.1;
                b_old = b;
                b = b > 0.9 ? 1.0 : b + 0.1;
            }
            if (p > L96) {
                a += (a_old - a) / 2.0;
                b -= (b - b_old) / 2.0;
            }
            p = ln_simpson_dev(g, r, a, b, 11);
            if (p >= L95 && p < L96) break;
        }
    } else {
        double a_r = double(ml) * 0.01, a_l = 0.0, b_r = 1.0, b_l = double(ml) * 0.01;
        for (int counter = 0; counter != 10; counter++) {
            if (p < L95) {
                a_r = a;
                a = a < 0.1 ? 0.0 : a - ((a - a_l) / 2.0);
                b_l = b;
                b = b > 0.9 ? 1.0 : b + ((b_r - b) / 2.0);
            }
            if (p > L96) {
                a_l = a;
                a += (a_r - a) / 2.0;
                b_r = b;
                b -= (b - b_l) / 2.0;
            }
            p = ln_simpson_dev(g, r, a, b, 11);
            if (p >= L95 && p < L96) break;
        }
    }
    ci.ml = ml;
    ci.a = a;
    ci.b = b;
    out[gi] = ci;
}

void device_credible_intervals(const uint64_t* d_grp_off, const uint8_t* d_grp_final, const double* d_alt, const uint32_t* d_depth,
                               uint64_t n_groups, const double* d_ln_fact, CredibleInterval* d_out, hipStream_t stream) {
    if (!n_groups) return;
    dim3 grid(uint32_t((n_groups + 63) / 64)), block(64);
    hipLaunchKernelGGL(k6_credible_intervals, grid, block, 0, stream, d_grp_off, d_grp_final, d_alt, d_depth, n_groups, d_ln_fact, d_out);
    HIP_OK_(hipGetLastError());
}

}  // namespace mp
