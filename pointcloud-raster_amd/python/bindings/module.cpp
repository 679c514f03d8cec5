// module.cpp -- entry point of pcr._pcr: the reference's Python surface (python/bindings.cpp)
// over the MI355X engine.  Registration is split by layer: bind_core / bind_engine / bind_io.
#include "common.h"

PYBIND11_MODULE(_pcr, m) {
    m.doc() = "Point Cloud Reduction -- MI355X (gfx950) HIP engine behind the pcr API";
    bind_core(m);
    bind_engine(m);
    bind_io(m);
}
