// Stand-in of the diagnostic CPU build (tools/ref_oracle/README.md); written for this repository.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>
namespace thrust {
struct device_t {}; static const device_t device{};
template <class T> struct plus { T operator()(const T& a, const T& b) const { return a + b; } };
template <class K, class V> void sort_by_key(device_t, K* kb, K* ke, V* v) {
  size_t n = ke - kb; std::vector<size_t> idx(n); std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return kb[a] < kb[b]; });
  std::vector<K> k2(n); std::vector<V> v2(n);
  for (size_t i = 0; i < n; ++i) { k2[i] = kb[idx[i]]; v2[i] = v[idx[i]]; }
  for (size_t i = 0; i < n; ++i) { kb[i] = k2[i]; v[i] = v2[i]; }
}
template <class T> void exclusive_scan(device_t, T* b, T* e, T* o) { T run = 0; for (; b != e; ++b, ++o) { T v = *b; *o = run; run += v; } }
template <class T, class U> void fill(device_t, T* b, T* e, U v) { for (; b != e; ++b) *b = (T)v; }
template <class I, class O, class F> void transform(device_t, I b, I e, O o, F f) { for (; b != e; ++b, ++o) *o = f(*b); }
template <class I, class J, class O, class F> void transform(device_t, I b, I e, J c, O o, F f) { for (; b != e; ++b, ++c, ++o) *o = f(*b, *c); }
template <class I, class T, class F> T reduce(device_t, I b, I e, T init, F f) { for (; b != e; ++b) init = f(init, *b); return init; }
}
