// Differential driver for ba::InterpolationBufferT: builds a few deterministic buffers, runs a
// fixed list of in-range queries (GetElement with index, GetNext walks, GetRange) and prints the
// results as JSON with 17 significant digits.  The SAME source is compiled twice:
//   * tests/golden/make_interp_golden.py compiles it against the reference's header
//     (-I/root/reference/include, build container only) and commits the output as
//     tests/golden/interp_buffer.json;
//   * tests/test_interpolation_buffer.py compiles it against include/ba/InterpolationBuffer.h
//     and compares the output with that file, value for value.
// Queries stay inside [start_time, end_time]: outside it the reference converts a negative
// double to size_t (undefined behaviour), and its own documentation asks callers to check
// HasElement first.
#include <algorithm>
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <ba/InterpolationBuffer.h>

struct Sample {
  double a, b, time;
  Sample operator*(double s) const { return {a * s, b * s, time}; }
  Sample operator+(const Sample& o) const { return {a + o.a, b + o.b, time}; }
};
typedef ba::InterpolationBufferT<Sample, double> Buffer;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand() {  // splitmix64 -> [0,1)
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

static void print_sample(const Sample& s) { std::printf("[%.17g, %.17g, %.17g]", s.a, s.b, s.time); }

static void dump_buffer(const char* name, const Buffer& buf, const std::vector<double>& qs,
                        const std::vector<std::pair<double, double>>& ranges, bool last) {
  std::printf(" \"%s\": {\n  \"n\": %zu, \"start_time\": %.17g, \"end_time\": %.17g, \"average_dt\": %.17g,\n",
              name, buf.elements.size(), buf.start_time, buf.end_time, buf.average_dt);
  std::printf("  \"get_element\": [\n");
  for (size_t i = 0; i < qs.size(); ++i) {
    size_t idx = 12345;
    const Sample s = buf.GetElement(qs[i], &idx);
    std::printf("   {\"t\": %.17g, \"has\": %d, \"index\": %zu, \"value\": ", qs[i], buf.HasElement(qs[i]) ? 1 : 0, idx);
    print_sample(s);
    std::printf("}%s\n", i + 1 < qs.size() ? "," : "");
  }
  std::printf("  ],\n  \"get_range\": [\n");
  for (size_t i = 0; i < ranges.size(); ++i) {
    const std::vector<Sample> r = buf.GetRange(ranges[i].first, ranges[i].second);
    std::printf("   {\"start\": %.17g, \"end\": %.17g, \"elements\": [", ranges[i].first, ranges[i].second);
    for (size_t k = 0; k < r.size(); ++k) {
      print_sample(r[k]);
      if (k + 1 < r.size()) std::printf(", ");
    }
    std::printf("]}%s\n", i + 1 < ranges.size() ? "," : "");
  }
  std::printf("  ],\n  \"get_next\": [\n");
  // walks: from the element found at a start time, GetNext until it reports the end
  for (size_t i = 0; i < ranges.size(); ++i) {
    double t0 = std::max(ranges[i].first, buf.start_time), t1 = std::min(ranges[i].second, buf.end_time);
    size_t idx = 0;
    (void)buf.GetElement(t0, &idx);
    std::printf("   {\"from\": %.17g, \"max_time\": %.17g, \"steps\": [", t0, t1);
    Sample m{0, 0, 0};
    int guard = 0;
    bool more = true;
    while (more && guard++ < 100000) {
      more = buf.GetNext(t1, idx, m);
      std::printf("%s{\"ret\": %d, \"index\": %zu, \"value\": ", guard > 1 ? ", " : "", more ? 1 : 0, idx);
      print_sample(m);
      std::printf("}");
    }
    std::printf("]}%s\n", i + 1 < ranges.size() ? "," : "");
  }
  std::printf("  ]\n }%s\n", last ? "" : ",");
}

int main() {
  std::printf("{\n");
  for (int which = 0; which < 4; ++which) {
    Buffer buf;
    std::vector<double> times;
    const char* name;
    if (which == 0) {          // uniform 100 Hz, 400 samples (the shape of an IMU stream)
      name = "uniform_100hz";
      for (int i = 0; i < 400; ++i) times.push_back(10.0 + 0.01 * i);
    } else if (which == 1) {   // jittered sampling intervals
      name = "jittered";
      double t = 3.25;
      for (int i = 0; i < 300; ++i) { times.push_back(t); t += 0.004 + 0.012 * urand(); }
    } else if (which == 2) {   // a long gap in the middle: the index guess is far off
      name = "gap";
      double t = 0.0;
      for (int i = 0; i < 200; ++i) { times.push_back(t); t += (i == 100) ? 5.0 : 0.01; }
    } else {                   // two samples only
      name = "two_samples";
      times.push_back(1.0); times.push_back(1.5);
    }
    for (size_t i = 0; i < times.size(); ++i) {
      const double t = times[i];
      buf.AddElement({0.5 * t * t - 3.0 * t, 1.0 + (double)i, t});
    }
    std::vector<double> qs;
    std::vector<std::pair<double, double>> ranges;
    const double t0 = times.front(), t1 = times.back();
    qs.push_back(t0); qs.push_back(t1);
    for (int i = 0; i < 120; ++i) qs.push_back(t0 + (t1 - t0) * urand());
    for (int i = 0; i < 60; ++i) qs.push_back(times[(size_t)(urand() * times.size()) % times.size()]);  // coincident
    for (int i = 0; i < 40; ++i) {
      double a = t0 + (t1 - t0) * urand(), b = t0 + (t1 - t0) * urand();
      if (a > b) std::swap(a, b);
      if (which != 3 && b - a > 0.05 * (t1 - t0)) b = a + 0.05 * (t1 - t0) * urand();
      ranges.push_back({a, b});
    }
    for (int i = 0; i < 20; ++i) {  // pose times that coincide with stored samples (both ends, one end)
      const size_t ia = (size_t)(urand() * times.size()) % times.size();
      const size_t ib = std::min(times.size() - 1, ia + 1 + (size_t)(urand() * 20));
      ranges.push_back({times[ia], times[ib]});
      ranges.push_back({times[ia], std::min(t1, times[ib] + 0.003)});
    }
    ranges.push_back({t0 - 1.0, t1 + 1.0});  // trimmed to the covered span
    ranges.push_back({t0, t0});
    ranges.push_back({t1, t1});
    dump_buffer(name, buf, qs, ranges, which == 3);
  }
  std::printf("}\n");
  return 0;
}
