// solver.hpp -- source-compatibility forwarder: the reference's header of this name is provided by
// mi355cg_compat.hpp (classes re-implemented over the MI355X C ABI, include/mi355cg.h).
#pragma once
#include "mi355cg_compat.hpp"
