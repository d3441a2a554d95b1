// Sanitizer target of the CPU port (TEST INFRASTRUCTURE): reads a system written by tests/test_cpu_port.py, evaluates
// energies and forces, runs a few Langevin steps with list rebuilds, prints the numbers.  Built twice by
// oracle/Makefile: plain, and with -fsanitize=address,undefined (the test compares the two outputs and requires a
// clean sanitizer log).  File layout: int32 {model, n, n_bonded, n_params, has_box, n_steps}, then seq int32[n],
// is_end int32[n], bonded int32[n_bonded][2], box double[3], flat double[n_params], center double[n][3], quat double[n][4],
// and for model 4 (oxNA; n_params = three vectors) is_rna int32[n].
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
void* mythos_cpu_create(int model, int n, const int32_t* seq, const uint8_t* is_end, int n_bonded, const int32_t* bonded,
                        const double* box, const double* flat, int n_params);
void mythos_cpu_destroy(void* h);
void mythos_cpu_set_types(void* h, const uint8_t* is_rna);
int mythos_cpu_build_pairs(void* h, const double* center, double r_list);
int mythos_cpu_energy(void* h, const double* center, const double* quat, double* e_terms, double* dU_dcenter,
                      double* dU_dquat, double* torque_body);
int mythos_cpu_langevin_run(void* h, double* c, double* q, double* p, double* L, int n_steps, double dt, double kT,
                            double gamma_t, double gamma_r, double mass, const double* inertia, uint64_t seed,
                            int64_t step0, double r_cut, double skin, int rebuild_every, double* e_last);
}

template <typename T>
static std::vector<T> rd(FILE* f, size_t n) {
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
  return v;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  const auto hdr = rd<int32_t>(f, 6);
  const int model = hdr[0], n = hdr[1], nb = hdr[2], np = hdr[3], has_box = hdr[4], n_steps = hdr[5];
  const auto seq = rd<int32_t>(f, n);
  const auto end32 = rd<int32_t>(f, n);
  const auto bonded = rd<int32_t>(f, 2 * (size_t)nb);
  const auto box = rd<double>(f, 3);
  const auto flat = rd<double>(f, np);
  auto c = rd<double>(f, 3 * (size_t)n);
  auto q = rd<double>(f, 4 * (size_t)n);
  const auto rna32 = model == 4 ? rd<int32_t>(f, n) : std::vector<int32_t>();
  fclose(f);
  std::vector<uint8_t> is_end(end32.begin(), end32.end());
  void* h = mythos_cpu_create(model, n, seq.data(), is_end.data(), nb, bonded.data(), has_box ? box.data() : nullptr, flat.data(), np);
  if (!h) return 3;
  if (model == 4) {
    std::vector<uint8_t> is_rna(rna32.begin(), rna32.end());
    mythos_cpu_set_types(h, is_rna.data());
  }
  mythos_cpu_build_pairs(h, c.data(), 3.25);
  double e[8];
  std::vector<double> dc(3 * (size_t)n), dq(4 * (size_t)n), tb(3 * (size_t)n);
  mythos_cpu_energy(h, c.data(), q.data(), e, dc.data(), dq.data(), tb.data());
  printf("E");
  for (double v : e) printf(" %.12e", v);
  double fs[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) fs[k] += dc[3 * (size_t)i + k];
  printf("\nFSUM %.3e %.3e %.3e\n", fs[0], fs[1], fs[2]);
  std::vector<double> p(3 * (size_t)n, 0.0), L(3 * (size_t)n, 0.0);
  const double inertia[3] = {1, 1, 1};
  double el[10];
  const int builds = mythos_cpu_langevin_run(h, c.data(), q.data(), p.data(), L.data(), n_steps, 0.005, 0.0987166667, 0.0394866667,
                                             0.0131622222, 1.0, inertia, 42, 0, 3.25, 0.3, 5, el);
  printf("MD builds %d E", builds);
  for (double v : el) printf(" %.10e", v);
  printf("\nX %.12e %.12e %.12e\n", c[0], c[3 * (size_t)(n - 1) + 2], q[4 * (size_t)(n / 2)]);
  mythos_cpu_destroy(h);
  return 0;
}
