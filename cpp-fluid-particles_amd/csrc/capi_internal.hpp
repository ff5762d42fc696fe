// capi_internal.hpp — what the C-ABI translation units (capi.hip, slab.hip) share: the object behind the
// opaque sphx_system handle and a few helpers.
#pragma once

#include <memory>
#include <string>

#include "BasicSPHSolver.h"
#include "DFSPHSolver.h"
#include "PBDSolver.h"
#include "SPHSystem.h"
#include "sphx_c.h"

struct sphx_system {
    sphx_params params;
    std::unique_ptr<SPHSystem> system;
    BasicSPHSolver* wcsph = nullptr;   // non-owning views of the solver the system owns
    DFSPHSolver* dfsph = nullptr;
    PBDSolver* pbd = nullptr;
    int n = 0, nb = 0, cells = 0;
};

int sphx_fail(int code, const std::string& msg);      // records the text for sphx_last_error(), returns code
int sphx_create_impl(const sphx_params* P, const float* fluid, int n, const float* boundary, int nb, int run_ctor_step,
                     sphx_system** out);              // may throw (callers sit inside a guarded scope)
extern "C" int sphx_locate(const sphx_system* h, int field, void** ptr, size_t* bytes);   // device pointer + byte size of a field
