// standalone sanitizer driver: exact-size heap buffers, one small solve through every entry point
#include "lfsd_capi.cpp"   // the product C ABI translation unit, compiled with -DLFSD_EMU
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <stdlib.h>
int main() {
  lfsd_model_info mi; lfsd_get_model_info(&mi);
  const int N = getenv("LFSD_SAN_N") ? atoi(getenv("LFSD_SAN_N")) : 6;      // (>= 40: the wide kernel's multiple-shooting steps run)
  const int B = 5, n = mi.n_state, m = mi.n_control, p = mi.n_auxvar, nc = mi.n_const, nw = 2, ni = 1;
  for (int pass = 0; pass < 4; ++pass) {         // both arithmetic types x both mappings of the OC solve (lock-step, wide)
    const int dtype = pass & 1;
    const int mapping = (pass & 2) ? LFSD_MAP_WIDE : LFSD_MAP_LOCKSTEP;
    const size_t es = dtype ? 8 : 4;
    auto buf = [&](size_t cnt) { return std::vector<char>(cnt * es); };
    auto x0 = buf(B * n), hz = buf(B), th = buf(B * p), cs = buf(nc ? nc : 1), X = buf(B * (N + 1) * n), U = buf(B * (N + 1) * m),
         L = buf(B * (N + 1) * n), cost = buf(B), Z = buf((size_t)B * (N + 1) * (n + p) * n), taus = buf(B * nw), wps = buf(B * nw * ni),
         loss = buf(B), grad = buf(B * p), aX = buf((size_t)B * (N + 1) * p * n), aU = buf((size_t)B * (N + 1) * p * m),
         mth = buf(B * p), mm = buf(B * p), mv = buf(B * p), mvh = buf(B * p), la = buf(B * p);
    auto set = [&](std::vector<char>& v, size_t i, double val) { if (dtype) ((double*)v.data())[i] = val; else ((float*)v.data())[i] = (float)val; };
    for (int b = 0; b < B; ++b) { set(hz, b, 1.0); for (int i = 0; i < p; ++i) set(th, b * p + i, 1.0 + 0.1 * i + 0.05 * b);
      for (int w = 0; w < nw; ++w) { set(taus, b * nw + w, 0.3 + 0.4 * w); set(wps, b * nw + w, 0.5); } }
    for (int i = 0; i < nc; ++i) set(cs, i, lfsd_const_default(i));
    std::vector<int> it(B), st(B), iface = {0}, stats(4 * B);
    size_t wsb = lfsd_coc_workspace_bytes(dtype, B, N, 3, mapping, 0);
    std::vector<char> ws(wsb);
    int rc = lfsd_coc_solve(dtype, B, N, 4, x0.data(), hz.data(), th.data(), nc ? cs.data() : nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, X.data(), U.data(), L.data(),
                            cost.data(), it.data(), st.data(), 40, dtype ? 1e-9 : 1e-6, 3, mapping, ws.data(), wsb, nullptr);
    printf("dtype %d coc rc %d status %d iters %d\n", dtype, rc, st[0], it[0]);
    if (pass == 2) {
      // fp32 on the wide mapping: the launch schemes of lfsd_capi.cpp's coc_solve_t (batch 5 <= capacity runs four wavefronts per
      // trajectory from the start where the model has them) -- one wavefront only; two launches handed over by the counter with a
      // capacity of 2; two launches handed over at iteration 2 -- under the sanitizers, outputs compared byte for byte
      const char* envs[3][3] = {{"LFSD_WIDE_WAVES", "1", nullptr}, {"LFSD_WIDE_CAPACITY", "2", nullptr}, {"LFSD_WIDE_CAPACITY", "2", "LFSD_WIDE_SUSPEND_IT"}};
      for (int v = 0; v < 3; ++v) {
        auto X2 = buf(B * (N + 1) * n), U2 = buf(B * (N + 1) * m), L2 = buf(B * (N + 1) * n), cost2 = buf(B);
        std::vector<int> it2(B), st2(B);
        setenv(envs[v][0], envs[v][1], 1);
        if (envs[v][2]) setenv(envs[v][2], "2", 1);
        std::vector<char> ws2(wsb);
        const int rc2 = lfsd_coc_solve(dtype, B, N, 4, x0.data(), hz.data(), th.data(), nc ? cs.data() : nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0,
                                       X2.data(), U2.data(), L2.data(), cost2.data(), it2.data(), st2.data(), 40, 1e-6, 3, mapping, ws2.data(), wsb, nullptr);
        unsetenv("LFSD_WIDE_WAVES"); unsetenv("LFSD_WIDE_CAPACITY"); unsetenv("LFSD_WIDE_SUSPEND_IT");
        const bool same = rc2 == 0 && X2 == X && U2 == U && L2 == L && cost2 == cost && it2 == it && st2 == st;
        printf("launch scheme %d: code %d same %d\n", v, rc2, (int)same);
        if (!same) return 3;
      }
    }
    rc = lfsd_aux_solve(dtype, B, N, hz.data(), th.data(), nc ? cs.data() : nullptr, 0, X.data(), U.data(), L.data(), Z.data(), nw, ni, iface.data(),
                        taus.data(), wps.data(), loss.data(), grad.data(), aX.data(), aU.data(), (pass & 2) ? 1 : 4, (pass & 2) ? 1e-3 : 0.0, stats.data(),
                        st.data(), (1 << LFSD_ST_FAILED) | ((pass & 1) ? (1 << st[1]) : 0), nullptr);      // (odd passes: row 1 is skipped whatever its status)
    printf("dtype %d aux units %d / %d unmet %d / %d\n", dtype, stats[0], stats[2], stats[1], stats[3]);
    printf("dtype %d aux rc %d\n", dtype, rc);
    for (int meth = 0; meth < 5; ++meth)
      rc |= lfsd_optimizer_step(dtype, meth, B, p, 0, 0.01, 0.9, 0.9, 0.999, 1e-8, th.data(), grad.data(), mm.data(), mv.data(), mvh.data(), nullptr, nullptr, nullptr);
    rc |= lfsd_lookahead(dtype, (long long)B * p, 0.9, th.data(), mm.data(), la.data(), nullptr);
    printf("dtype %d opt rc %d\n", dtype, rc);
  }
  return 0;
}
