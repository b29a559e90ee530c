// curve25519_amd/csrc/engine.hip -- the engine as ONE translation unit: its four parts included one after the other
// (engine_common.cuh says which is which).  The library is built from the parts, compiled in parallel (curve25519_amd/build.py);
// this file is what the single-file tools compile -- `hipcc -S` for the ISA tools (tools/valu_issue.py, tools/isa_mix_report.py),
// tests/test_resources.py and tools/resource_usage.py for the per-kernel register / scratch remarks, tools/build_variants.sh for
// A/B builds -- and gives the same kernels.
#include "engine_x25519.hip"
#include "engine_fixed_base.hip"
#include "engine_verify.hip"
#include "engine_api.hip"
