// pcr/engine/filter.h -- point filter specification (API parity with the reference's
// include/pcr/engine/filter.h).  Both engines evaluate it before routing: a byte mask on the device
// (pcr_hip_filter_mask), a host loop in the host engine; a filtered-out point exists for no reduction.
#pragma once

#include "pcr/core/types.h"

#include <string>
#include <vector>

namespace pcr {

enum class CompareOp : uint8_t {
    Equal, NotEqual, Less, LessEqual, Greater, GreaterEqual, InSet, NotInSet
};

struct FilterPredicate {
    std::string channel_name;
    CompareOp op = CompareOp::Equal;
    float value = 0.0f;
    std::vector<float> value_set;
};

struct FilterSpec {
    std::vector<FilterPredicate> predicates;

    FilterSpec& add(const std::string& channel, CompareOp op, float value) {
        FilterPredicate p;
        p.channel_name = channel;
        p.op = op;
        p.value = value;
        predicates.push_back(p);
        return *this;
    }
    FilterSpec& add_in_set(const std::string& channel, const std::vector<float>& values) {
        FilterPredicate p;
        p.channel_name = channel;
        p.op = CompareOp::InSet;
        p.value_set = values;
        predicates.push_back(p);
        return *this;
    }
    bool empty() const { return predicates.empty(); }
};

}  // namespace pcr
