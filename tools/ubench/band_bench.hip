// Kernel-only bench of band.hip (diagnostic tool): CTC-shaped pairs, no host engine.
//   hipcc --offload-arch=gfx950 -O3 -w -I gtn_amd/csrc -I include [-DGTNX_BAND_TIMING] tools/ubench/band_bench.hip -o tools/ubench/band_bench
//   tools/ubench/band_bench [B T C U]
#include "../../gtn_amd/csrc/band.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace gtnx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#if !defined(__HIP_DEVICE_COMPILE__)
int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 512, T = argc > 2 ? atoi(argv[2]) : 1000, C = argc > 3 ? atoi(argv[3]) : 256,
      U = argc > 4 ? atoi(argv[4]) : 100;
  const int N = 2 * U + 1, npl = getenv("BAND_NPL") ? atoi(getenv("BAND_NPL")) : band_npl(N), NS = band_row_stride(N, npl);
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> ud(-5.f, 5.f);
  std::vector<float> em(size_t(B) * T * C);
  for (auto& v : em) v = ud(rng);
  float* d_em;
  CK(hipMalloc(&d_em, em.size() * 4));
  CK(hipMemcpy(d_em, em.data(), em.size() * 4, hipMemcpyHostToDevice));
  float* d_grad;
  const size_t grad_off = getenv("GRAD_OFF") ? size_t(atol(getenv("GRAD_OFF"))) : 0;  // bytes: where the gradient tensor starts relative to its allocation
  CK(hipMalloc(&d_grad, em.size() * 4 + grad_off + 256));
  d_grad = reinterpret_cast<float*>(reinterpret_cast<char*>(d_grad) + grad_off);
  std::vector<BandPair> pairs(B);
  size_t per_nodes = sizeof(BandNode) * N, per_flags = (N + 63) / 64 * 64, per_s = 4 * size_t(N);
  size_t per_g = per_nodes + per_flags + 2 * per_s;
  char* d_g;
  CK(hipMalloc(&d_g, per_g * B));
  std::vector<char> h_g(per_g * B);
  const size_t per_alpha = size_t(T + 1) * NS * 4, per_off = 8 * (4 * size_t(T) + 16);
  char *d_alpha, *d_off;
  CK(hipMalloc(&d_alpha, per_alpha * B));
  CK(hipMalloc(&d_off, per_off * B));
  float *d_score, *d_delta, *d_gfix, *d_norm, *d_rowlse;
  CK(hipMalloc(&d_score, 4 * B));
  CK(hipMalloc(&d_norm, 4 * B));
  CK(hipMalloc(&d_rowlse, 4 * size_t(B) * T));
  CK(hipMalloc(&d_delta, 4 * B));
  CK(hipMalloc(&d_gfix, 4 * size_t(B) * 3 * N));
  CK(hipMemset(d_gfix, 0, 4 * size_t(B) * 3 * N));
  std::vector<float> ones(B, -1.0f);
  CK(hipMemcpy(d_delta, ones.data(), 4 * B, hipMemcpyHostToDevice));
  int A = 0;
  for (int b = 0; b < B; ++b) {
    std::vector<int> tg(U);
    for (auto& v : tg) v = 1 + rng() % (C - 1);
    BandNode* nd = reinterpret_cast<BandNode*>(h_g.data() + per_g * b);
    uint8_t* fl = reinterpret_cast<uint8_t*>(h_g.data() + per_g * b + per_nodes);
    int* sn = reinterpret_cast<int*>(h_g.data() + per_g * b + per_nodes + per_flags);
    int* sl = sn + N;
    int a = 0;
    std::vector<std::pair<int, int>> ln;
    for (int m = 0; m < N; ++m) {
      const int lab = m % 2 ? tg[(m - 1) / 2] : 0;
      nd[m].lab = lab;
      nd[m].aid[0] = a++;
      nd[m].aid[1] = m > 0 ? a++ : -1;
      nd[m].aid[2] = (m % 2 && m > 1 && lab != tg[(m - 1) / 2 - 1]) ? a++ : -1;
      fl[m] = (m == 0 ? NF_START : 0) | (m >= N - 2 ? NF_ACCEPT : 0);
      ln.push_back({lab, m});
    }
    A = a;
    std::sort(ln.begin(), ln.end());
    for (int i = 0; i < N; ++i) {
      sl[i] = ln[i].first;
      sn[i] = ln[i].second;
    }
    BandPair& p = pairs[b];
    p = BandPair{};
    p.nodes = reinterpret_cast<BandNode*>(d_g + per_g * b);
    p.nflags = reinterpret_cast<uint8_t*>(d_g + per_g * b + per_nodes);
    p.snode = reinterpret_cast<int*>(d_g + per_g * b + per_nodes + per_flags);
    p.slab = p.snode + N;
    p.n_lab = N;
    p.w = nullptr;
    p.em = d_em + size_t(b) * T * C;
    p.em_copy = getenv("EMCOPY") ? d_grad + size_t(b) * T * C : nullptr;  // (the copy variant of the forward sweep; d_grad is rewritten by the backward sweep afterwards)
    p.alpha = reinterpret_cast<float*>(d_alpha + per_alpha * b);
    p.aoff = reinterpret_cast<double*>(d_off + per_off * b);
    p.score = d_score + b;
    p.norm = getenv("FUSE") ? d_norm + b : nullptr;
    p.rowlse = getenv("FUSE") ? d_rowlse + size_t(b) * T : nullptr;
    p.delta = d_delta + b;
    p.delta_norm = getenv("FUSE") ? d_delta + b : nullptr;
    p.grad_em = d_grad + size_t(b) * T * C;
    p.grad_fixed = getenv("NOGRADG") ? nullptr : d_gfix + size_t(b) * 3 * N;
    p.N = N;
    p.T = T;
    p.C = C;
    p.NS = NS;
    p.hot = 0;
    p.lgrn = band_forward_lgrn(C);
  }
  CK(hipMemcpy(d_g, h_g.data(), h_g.size(), hipMemcpyHostToDevice));
  BandPair* d_pairs;
  CK(hipMalloc(&d_pairs, sizeof(BandPair) * B));
  CK(hipMemcpy(d_pairs, pairs.data(), sizeof(BandPair) * B, hipMemcpyHostToDevice));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&e2));
  const bool gradg = getenv("NOGRADG") == nullptr;
  float fsum = 0, bsum = 0;
  const int reps = 20;
  for (int it = 0; it < reps + 3; ++it) {
    CK(hipEventRecord(e0, 0));
    launch_band_forward(d_pairs, B, npl, C, NS, true, C % 4 == 0, 0);
    CK(hipEventRecord(e1, 0));
    launch_band_backward(d_pairs, B, npl, C, NS, true, gradg, C % 4 == 0, 0);
    CK(hipEventRecord(e2, 0));
    CK(hipEventSynchronize(e2));
    float f, bw;
    CK(hipEventElapsedTime(&f, e0, e1));
    CK(hipEventElapsedTime(&bw, e1, e2));
    if (it >= 3) {
      fsum += f;
      bsum += bw;
    }
  }
  const double fb = double(B) * (4.0 * T * C + 4.0 * (T + 1) * NS), bb = double(B) * (8.0 * T * C + 4.0 * (T + 1) * NS);
  printf("B %d T %d C %d U %d (N %d NS %d npl %d Kf %d Kb %d): forward %.3f ms (%.2f TB/s)  backward %.3f ms (%.2f TB/s)\n", B, T,
         C, U, N, NS, npl, band_block_rows(C, 0, false), band_block_rows(C, NS, true), fsum / reps,
         fb / (fsum / reps) * 1e-9, bsum / reps, bb / (bsum / reps) * 1e-9);
  std::vector<float> sc(B);
  CK(hipMemcpy(sc.data(), d_score, 4 * B, hipMemcpyDeviceToHost));
  printf("score[0..3] %.4f %.4f %.4f %.4f\n", sc[0], sc[1], sc[2], sc[3]);
  // ---- utterances 0 and B-1 against the same recursion in float64 on the host (delta = -1, FUSE: + softmax term)
  for (int b : {0, B - 1}) {
    const BandNode* nd = reinterpret_cast<const BandNode*>(h_g.data() + per_g * b);
    const float* e = em.data() + size_t(b) * T * C;
    const double NEG = -1e300;
    auto lse = [&](double a, double c) { const double m = a > c ? a : c; return m <= NEG ? NEG : m + std::log(std::exp(a - m) + std::exp(c - m)); };
    std::vector<double> al(size_t(T + 1) * N, NEG), be(size_t(T + 1) * N, NEG);
    al[0] = 0.0;
    for (int t = 0; t < T; ++t)
      for (int m = 0; m < N; ++m) {
        double v = al[size_t(t) * N + m];
        if (m > 0) v = lse(v, al[size_t(t) * N + m - 1]);
        if (nd[m].aid[2] >= 0) v = lse(v, al[size_t(t) * N + m - 2]);
        al[size_t(t + 1) * N + m] = v <= NEG ? NEG : v + e[size_t(t) * C + nd[m].lab];
      }
    be[size_t(T) * N + N - 1] = 0.0;
    if (N > 1) be[size_t(T) * N + N - 2] = 0.0;
    for (int t = T - 1; t >= 0; --t)
      for (int m = 0; m < N; ++m) {
        auto q = [&](int k) { const double x = be[size_t(t + 1) * N + k]; return x <= NEG ? NEG : x + e[size_t(t) * C + nd[k].lab]; };
        double v = q(m);
        if (m + 1 < N) v = lse(v, q(m + 1));
        if (m + 2 < N && nd[m + 2].aid[2] >= 0) v = lse(v, q(m + 2));
        be[size_t(t) * N + m] = v;
      }
    const double z = lse(al[size_t(T) * N + N - 1], N > 1 ? al[size_t(T) * N + N - 2] : NEG);
    std::vector<double> g(size_t(T) * C, 0.0), ga(3 * size_t(N), 0.0);
    const bool fuse = getenv("FUSE") != nullptr;
    for (int t = 0; t < T; ++t) {
      if (fuse) {
        double mx = NEG, s = 0;
        for (int c = 0; c < C; ++c) mx = e[size_t(t) * C + c] > mx ? e[size_t(t) * C + c] : mx;
        for (int c = 0; c < C; ++c) s += std::exp(e[size_t(t) * C + c] - mx);
        for (int c = 0; c < C; ++c) g[size_t(t) * C + c] = -std::exp(e[size_t(t) * C + c] - mx) / s;  // delta_norm = -1
      }
      for (int m = 0; m < N; ++m) {
        g[size_t(t) * C + nd[m].lab] -= std::exp(al[size_t(t + 1) * N + m] + be[size_t(t + 1) * N + m] - z);  // delta = -1
        for (int k = 0; k < 3; ++k)
          if (nd[m].aid[k] >= 0 && m - k >= 0)
            ga[nd[m].aid[k]] -= std::exp(al[size_t(t) * N + m - k] + e[size_t(t) * C + nd[m].lab] + be[size_t(t + 1) * N + m] - z);
      }
    }
    std::vector<float> hg(size_t(T) * C), hga(3 * size_t(N));
    CK(hipMemcpy(hg.data(), d_grad + size_t(b) * T * C, hg.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hga.data(), d_gfix + size_t(b) * 3 * N, hga.size() * 4, hipMemcpyDeviceToHost));
    double ge = 0, gae = 0;
    int worst = -1;
    for (size_t i = 0; i < hg.size(); ++i) ge = std::max(ge, std::fabs(double(hg[i]) - g[i]));
    if (gradg)
      for (int a = 0; a < A; ++a) {
        const double r = std::fabs(double(hga[a]) - ga[a]) / std::max(1.0, std::fabs(ga[a]));
        if (r > gae) gae = r, worst = a;
      }
    if (worst >= 0) printf("  worst arc %d: gpu %.6f float64 %.6f\n", worst, double(hga[worst]), ga[worst]);
    printf("utterance %d against float64: score %.6f (gpu %.6f), max |d emission| error %.3g, max relative target-arc gradient error %.3g\n", b, z,
           double(sc[b]), ge, gae);
  }
#ifdef GTNX_BAND_TIMING
  long long h[256];
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(::g_band_timing), sizeof(h)));
  for (int k = 0; k < 2; ++k) {
    printf("%s timing, workgroup 0, cycles per tick (loop | land | issue | store/drain | lse | compute | barrier), ticks %lld\n",
           k ? "backward" : "forward", h[k * 128 + 127]);
    for (int wv = 0; wv < (k ? 12 : 8); ++wv) {
      printf("  wave %d:", wv);
      for (int i = 0; i < 7; ++i) printf(" %6.0f", double(h[k * 128 + wv * 7 + i]) / double(h[k * 128 + 127] ? h[k * 128 + 127] : 1));
      printf("\n");
    }
  }
#endif
  return 0;
}
#endif
