// Kernel-only bench of band.hip's Viterbi kernels (diagnostic tool): CTC-shaped pairs, no host engine.
//   hipcc --offload-arch=gfx950 -O3 -w -I gtn_amd/csrc -I include [-DGTNX_VIT_NO_CHASE] tools/ubench/viterbi_bench.hip -o tools/ubench/viterbi_bench
//   tools/ubench/viterbi_bench [B T C U]
#include "../../gtn_amd/csrc/band.hip"

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
  const int N = 2 * U + 1, NS = band_row_stride(N, band_npl(N));
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> ud(-5.f, 5.f);
  std::vector<float> em(size_t(B) * T * C);
  for (auto& v : em) v = ud(rng);
  float* d_em;
  CK(hipMalloc(&d_em, em.size() * 4));
  CK(hipMemcpy(d_em, em.data(), em.size() * 4, hipMemcpyHostToDevice));
  const size_t per_nodes = sizeof(BandNode) * N, per_flags = (N + 63) / 64 * 64, per_g = per_nodes + per_flags;
  std::vector<char> h_g(per_g * B);
  for (int b = 0; b < B; ++b) {
    std::vector<int> tg(U);
    for (auto& v : tg) v = 1 + rng() % (C - 1);
    BandNode* nd = reinterpret_cast<BandNode*>(h_g.data() + per_g * b);
    uint8_t* fl = reinterpret_cast<uint8_t*>(h_g.data() + per_g * b + per_nodes);
    int a = 0;
    for (int m = 0; m < N; ++m) {
      const int lab = m % 2 ? tg[(m - 1) / 2] : 0;
      nd[m].lab = lab;
      nd[m].aid[0] = a++;
      nd[m].aid[1] = m > 0 ? a++ : -1;
      nd[m].aid[2] = (m % 2 && m > 1 && lab != tg[(m - 1) / 2 - 1]) ? a++ : -1;
      fl[m] = uint8_t((m == 0 ? NF_START : 0) | ((m == N - 1 || m == N - 2) ? NF_ACCEPT : 0));
    }
  }
  char* d_g;
  CK(hipMalloc(&d_g, per_g * B));
  CK(hipMemcpy(d_g, h_g.data(), per_g * B, hipMemcpyHostToDevice));
  const size_t per_bp = (size_t(T) * NS + 512 + 255) / 256 * 256, per_pn = (4 * size_t(T + 1) + 255) / 256 * 256,
               per_pa = (12 * size_t(T) + 255) / 256 * 256, per = per_bp + per_pn + per_pa + 256;
  char* d_out;
  CK(hipMalloc(&d_out, per * B));
  std::vector<BandDecode> tab(B);
  for (int b = 0; b < B; ++b) {
    BandDecode& p = tab[b];
    p = BandDecode{};
    char* base = d_out + per * b;
    p.nodes = reinterpret_cast<const BandNode*>(d_g + per_g * b);
    p.nflags = reinterpret_cast<const uint8_t*>(d_g + per_g * b + per_nodes);
    p.w = nullptr;
    p.em = d_em + size_t(b) * T * C;
    p.bp = reinterpret_cast<uint8_t*>(base);
    p.pnode = reinterpret_cast<int*>(base + per_bp);
    p.path_arc = reinterpret_cast<int*>(base + per_bp + per_pn);
    p.path_lab = p.path_arc + T;
    p.path_w = reinterpret_cast<float*>(p.path_lab + T);
    p.path_len = reinterpret_cast<int*>(base + per_bp + per_pn + per_pa);
    p.score = reinterpret_cast<float*>(p.path_len + 1);
    p.tie = p.path_len + 2;
    p.N = N, p.T = T, p.C = C, p.NS = NS;
    p.stage_floats = 4096;
  }
  BandDecode* d_tab;
  CK(hipMalloc(&d_tab, sizeof(BandDecode) * B));
  CK(hipMemcpy(d_tab, tab.data(), sizeof(BandDecode) * B, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int wg = 0; wg < 2; ++wg) {
    // (launch_band_viterbi reads the switch once: call the kernels directly)
    auto launch = [&]() {
      if (false) {
      } else if (wg) {
        big_lds(band_viterbi_kernel);
        hipLaunchKernelGGL(band_viterbi_kernel, dim3(B), dim3(512), 4 * size_t(1032 + 4096) + 64, 0, d_tab);
      } else if (N <= 64) hipLaunchKernelGGL((band_viterbi_wave_kernel<1, false>), dim3(B), dim3(64), 0, 0, d_tab);
      else if (N <= 128) hipLaunchKernelGGL((band_viterbi_wave_kernel<2, false>), dim3(B), dim3(64), 0, 0, d_tab);
      else if (N <= 256) hipLaunchKernelGGL((band_viterbi_wave_kernel<4, false>), dim3(B), dim3(64), 0, 0, d_tab);
      else hipLaunchKernelGGL((band_viterbi_wave_kernel<8, false>), dim3(B), dim3(64), 0, 0, d_tab);
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double bytes = double(B) * (4.0 * T * C + 0.5 * T * N + 20.0 * T);
    float sc;
    int tie;
    CK(hipMemcpy(&sc, tab[0].score, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&tie, tab[0].tie, 4, hipMemcpyDeviceToHost));
    printf("%s: %.4f ms per launch  %.2f TB/s = %.1f %% of 8 TB/s   (score[0] %.4f tie[0] %d)\n", wg ? "workgroup kernel" : "one-wave kernel ", ms,
           bytes / (ms * 1e-3) / 1e12, 100.0 * bytes / (ms * 1e-3) / 8e12, sc, tie);
  }
  return 0;
}
#endif
