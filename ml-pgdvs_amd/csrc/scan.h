// Internal interface of the ordered stream compaction (scan.hip).
#pragma once
#include "common.h"

namespace pgdvs {
int64_t compact_workspace_bytes(int64_t n);
int compact_u8(const uint8_t *flags, int64_t n, int32_t *idx_out, int32_t *count_out,
               void *workspace, int64_t workspace_bytes, hipStream_t stream);
}  // namespace pgdvs
