// Umbrella header of the drop-in (mirrors the role of the reference's include/lexls/lexls.h):
//   LexLS::internal::LexLSE  — equality solver, HIP-backed (lexlse.h)
//   LexLS::internal::LexLSI  — active-set driver kept on the host (lexlsi.h), instantiated over the HIP-backed LexLSE
//   LexLS::LexLSE            — the reference's thin public wrapper
#pragma once

#include <lexls/lexlse.h>
#include <lexls/lexlsi.h>

namespace LexLS
{
    namespace internal
    {
        typedef LexLSI_T<LexLSE> LexLSI;
    }
} // namespace LexLS
