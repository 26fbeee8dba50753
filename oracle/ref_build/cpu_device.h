// TEST INFRASTRUCTURE ONLY (oracle/_ref build). Not part of the product.
//
// The reference hard-codes torch::kCUDA (src/solver.cpp:10, src/ulbm.cpp:42-88,
// src/differential.hpp:17, src/domain.cpp:7-11).  There is no GPU in the build
// container and the GPU box must time the reference on its *host* cores, so the
// unmodified reference sources are compiled with this header force-included
// (-include): libtorch is pulled in first (its own constexpr kCUDA definition is
// parsed untouched and header guards keep it from being re-parsed), then every
// later spelling of kCUDA in the reference sources resolves to kCPU.
// No reference source is copied or edited; see oracle/ref_build/Makefile.
#pragma once
#include <torch/torch.h>
#include <c10/core/DeviceType.h>
#define kCUDA kCPU
