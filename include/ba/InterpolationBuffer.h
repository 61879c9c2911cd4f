// ba::InterpolationBufferT — time-ordered sample buffer with linear interpolation, the
// pre-processing step in front of AddImuResidual (GetRange(t_i, t_j) yields the IMU
// measurements between two poses, end points interpolated).  Same public surface as the
// reference's /root/reference/include/ba/InterpolationBuffer.h:37-211 (members `elements`,
// `start_time`, `end_time`, `average_dt`; Clear / AddElement / GetNext / HasElement /
// GetElement / GetRange), written from its documented behaviour; off the GPU hot path
// (SURVEY.md §8f row 2).
//
// ElementType must provide `time`, `operator*(ScalarType)` and `operator+(ElementType)`.
//
// Behaviour is pinned against the reference's own header (it compiles stand-alone): the outputs
// of /root/reference/include/ba/InterpolationBuffer.h for a few hundred queries are committed as
// tests/golden/interp_buffer.json and this header must reproduce them bit for bit, including the
// index it reports when a query time coincides with a stored sample (the bracketing interval is
// found from an index guess, time / average interval, followed by a walk; a coinciding sample k
// is reported as interval k-1 with weight 1 when the guess started left of it, and GetRange
// then emits that sample twice — :147-191).  Inside [start_time, end_time] only: outside it the
// reference's index guess is a negative double converted to size_t (undefined); callers are told
// to check HasElement first (:118-121), and GetRange trims to the covered span.
#ifndef BA_AMD_INTERPOLATION_BUFFER_H
#define BA_AMD_INTERPOLATION_BUFFER_H

#include <algorithm>
#include <cassert>
#include <cstddef>
#include <vector>

namespace ba {

template <typename ElementType, typename ScalarType>
struct InterpolationBufferT {
  std::vector<ElementType> elements;
  ScalarType start_time;
  ScalarType end_time;
  ScalarType average_dt;

  explicit InterpolationBufferT(unsigned int size = 1000) : start_time(-1), end_time(-1), average_dt(-1) {
    elements.reserve(size);
  }

  void Clear() {
    start_time = end_time = average_dt = -1;
    elements.clear();
  }

  // appends a sample (times strictly increasing), keeps start/end time and the running mean
  // of the sampling interval (the first sample contributes a zero interval, as in the reference)
  void AddElement(const ElementType& element) {
    assert(element.time > end_time);
    const std::size_t n = elements.size();
    const ScalarType dt = n == 0 ? ScalarType(0) : ScalarType(element.time - elements.back().time);
    average_dt = (average_dt == ScalarType(-1)) ? dt : (average_dt * ScalarType(n) + dt) / ScalarType(n + 1);
    elements.push_back(element);
    end_time = element.time;
    start_time = elements.front().time;
  }

  bool HasElement(const ScalarType time) const { return time >= start_time && time <= end_time; }

  ElementType GetElement(const ScalarType time) const {
    std::size_t index;
    return GetElement(time, &index);
  }

  // value at `time`; *index = left end of the interval the value was interpolated in.
  // Search as the reference does it (:147-191): start at floor((time - start_time) / average_dt),
  // clamped to the stored range, then walk towards `time`.  A walk from the right stops at the
  // first interval whose left sample is <= time, a walk from the left at the first interval whose
  // right sample is >= time — the two differ exactly when `time` is a stored sample.
  ElementType GetElement(const ScalarType time, std::size_t* index) const {
    assert(!elements.empty());
    const std::size_t n = elements.size();
    const ScalarType pos = (time - start_time) / average_dt;
    std::size_t k = pos > ScalarType(0) ? (std::size_t)pos : 0;  // (NaN / negative -> 0: out-of-span input)
    if (k > n - 1) k = n - 1;
    std::size_t left;  // interpolate between elements[left] and elements[left + 1]
    if (elements[k].time > time) {
      if (k == 0) { *index = 0; return elements.front(); }
      while (k > 1 && elements[k - 1].time > time) --k;
      left = k - 1;
    } else {
      if (k == n - 1) { *index = k; return elements.back(); }
      while (k + 1 < n && elements[k + 1].time < time) ++k;
      if (k == n - 1) { *index = k; return elements.back(); }  // time beyond the last sample
      left = k;
    }
    *index = left;
    const ElementType& e0 = elements[left];
    const ElementType& e1 = elements[left + 1];
    const ScalarType u = (time - e0.time) / (e1.time - e0.time);
    ElementType res = e0 * (1 - u) + e1 * u;
    res.time = time;
    return res;
  }

  // steps to the next stored sample not later than max_time (returns true), or interpolates
  // at max_time and returns false: the end of the walk
  bool GetNext(const ScalarType max_time, std::size_t& index_out, ElementType& output) const {
    if (index_out + 1 >= elements.size() || elements[index_out + 1].time > max_time) {
      output = GetElement(max_time, &index_out);
      return false;
    }
    ++index_out;
    output = elements[index_out];
    return true;
  }

  // all samples of [start, end] (trimmed to the covered span), both end points interpolated
  std::vector<ElementType> GetRange(ScalarType start, ScalarType end) const {
    std::vector<ElementType> out;
    if (start < start_time) start = start_time;
    if (end > end_time) end = end_time;
    if (!elements.empty() && HasElement(start)) {
      std::size_t index;
      out.push_back(GetElement(start, &index));
      ElementType m;
      while (GetNext(end, index, m)) out.push_back(m);
      out.push_back(m);
    }
    return out;
  }
};

}  // namespace ba

#endif  // BA_AMD_INTERPOLATION_BUFFER_H
