// engine_internal.h -- shared by the HIP translation units of the engine (engine.hip, encode.hip); include after hip_runtime.h
#pragma once
#include "engine.h"

namespace anx {

// per-device scratch pool (engine.hip): freed blocks are kept for the next batch
hipError_t pool_malloc(void** p, size_t bytes);
void pool_free(void* p);
// device-side query encoder (encode.hip): fills the query and tile arrays of `b` from the packed inputs
int batch_encode_device(const HostModel& m, const DeviceLexicon* dl, Batch* b, const char* blob, const uint32_t* off, size_t n,
                        const anx_params& p, std::string& err);

}  // namespace anx
