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
// Difference from the reference: the element AT OR BEFORE a query time is found by binary
// search instead of an index guess from the average sampling interval followed by a walk.
// Results are identical except when a query time coincides exactly with a stored sample: the
// reference's answer then depends on which side its guess started from (it may report the
// previous index and make GetRange emit that sample twice, once as a zero-length interval);
// here the coinciding sample is always reported once.
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

  // value at `time` (clamped to the first / last sample outside the covered span);
  // *index = the stored sample at or before `time`
  ElementType GetElement(const ScalarType time, std::size_t* index) const {
    assert(!elements.empty());
    const std::size_t n = elements.size();
    if (!(time > elements.front().time)) { *index = 0; return elements.front(); }
    if (!(time < elements.back().time)) { *index = n - 1; return elements.back(); }
    // first sample strictly after `time`
    std::size_t lo = 0, hi = n - 1;  // invariant: elements[lo].time <= time < elements[hi].time
    while (hi - lo > 1) {
      const std::size_t mid = lo + (hi - lo) / 2;
      if (elements[mid].time <= time) lo = mid; else hi = mid;
    }
    *index = lo;
    const ScalarType u = ScalarType(time - elements[lo].time) / ScalarType(elements[hi].time - elements[lo].time);
    ElementType res = elements[lo] * (ScalarType(1) - u) + elements[hi] * u;
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
