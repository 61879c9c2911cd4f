// Host test of ba::InterpolationBufferT (no GPU): the worked example recorded in SURVEY.md
// §8c for the reference's header — 10 samples at 0.0, 0.1, .., 0.9 with value = 10 t,
// GetRange(0.15, 0.55) -> 6 elements, first 1.5 @ 0.15, last 5.5 @ 0.55 — plus clamping,
// HasElement, GetNext and the running average interval.
#include <ba/InterpolationBuffer.h>
#include <cmath>
#include <cstdio>
struct Sample {
  double v, time;
  Sample operator*(double s) const { return {v * s, time}; }
  Sample operator+(const Sample& o) const { return {v + o.v, time}; }
};
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)
int main() {
  ba::InterpolationBufferT<Sample, double> buf;
  CHECK(buf.start_time == -1 && buf.end_time == -1 && buf.average_dt == -1);
  for (int i = 0; i < 10; ++i) buf.AddElement({1.0 * i, 0.1 * i});
  CHECK(buf.elements.size() == 10 && buf.start_time == 0.0 && std::fabs(buf.end_time - 0.9) < 1e-15);
  CHECK(std::fabs(buf.average_dt - 0.09) < 1e-12);  // nine 0.1 s intervals and the initial zero, over ten
  auto r = buf.GetRange(0.15, 0.55);
  CHECK(r.size() == 6);
  CHECK(std::fabs(r.front().v - 1.5) < 1e-12 && r.front().time == 0.15);
  CHECK(std::fabs(r.back().v - 5.5) < 1e-12 && r.back().time == 0.55);
  for (int i = 1; i <= 4; ++i) CHECK(std::fabs(r[i].v - (1.0 + i)) < 1e-12);
  CHECK(buf.HasElement(0.0) && buf.HasElement(0.9) && !buf.HasElement(0.95) && !buf.HasElement(-0.01));
  std::size_t idx = 99;
  CHECK(buf.GetElement(-1.0, &idx).v == 0.0 && idx == 0);   // clamped to the first sample
  CHECK(buf.GetElement(5.0, &idx).v == 9.0 && idx == 9);    // clamped to the last one
  CHECK(std::fabs(buf.GetElement(0.3, &idx).v - 3.0) < 1e-12 && (idx == 3 || idx == 2));
  auto all = buf.GetRange(-10.0, 10.0);                       // trimmed to the covered span
  CHECK(all.size() >= 10 && all.front().v == 0.0 && all.back().v == 9.0);
  Sample m{0, 0};
  idx = 8;
  CHECK(buf.GetNext(0.95, idx, m) && idx == 9 && m.v == 9.0);
  CHECK(!buf.GetNext(0.95, idx, m));
  buf.Clear();
  CHECK(buf.elements.empty() && buf.start_time == -1);
  CHECK(buf.GetRange(0.0, 1.0).empty());
  std::printf("ok\n");
  return 0;
}
