// common.h -- shared helpers of the _pcr extension module.
#pragma once

#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>
#include <string>

#include "pcr/core/types.h"

namespace py = pybind11;

// A failed Status surfaces in Python as RuntimeError(message), as in the reference module.
inline void raise_if_error(const pcr::Status& s) {
    if (!s.ok()) throw std::runtime_error(s.message);
}

void bind_core(py::module_& m);      // enums + value types + GridConfig + Grid + PointCloud
void bind_engine(py::module_& m);    // filter, glyph, reductions, pipeline
void bind_io(py::module_& m);        // I/O names kept for import compatibility (not part of this build)
