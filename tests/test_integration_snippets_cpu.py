"""CPU: the C/C++ that INTEGRATION.md tells a maintainer to write compiles against include/pmp.h - the in-process hook file
itself (tools/vtm_build/pmp_hook.cpp, syntax-checked against stub VTM declarations when the reference is absent is NOT attempted:
the real build is tests/test_vtm_roundtrip_cpu.py) and a free-standing C translation unit that uses every constant the document
names, so a wrong enum (the round-1 `PMP_NET_CHROMA_Q` where `PMP_CHROMA` belongs) cannot come back unnoticed."""
import os
import re
import subprocess

from conftest import ROOT

SNIPPET = r'''
#include <string.h>
#include "pmp.h"
/* INTEGRATION.md section 3/4: hosts pass a COMPONENT (PMP_LUMA / PMP_CHROMA) to the inference calls and a NET id to the loaders */
int hook(pmp_ctx *ctx, int qp, const uint8_t *by, const uint8_t *bu, const uint8_t *bv, int64_t n, int frames, int h, int w,
         uint8_t *hor, uint8_t *ver, uint8_t *qt, int8_t *dire, uint8_t *H, uint8_t *V, uint8_t *Q, int8_t *D)
{
    int k, rc;
    for (k = 0; k < 2; k++) {
        if ((rc = pmp_load_weights_file(ctx, k ? PMP_NET_CHROMA_Q : PMP_NET_LUMA_Q, qp, k ? "Chroma_Q_22.pmpw" : "Luma_Q_22.pmpw"))) return rc;
        if ((rc = pmp_load_weights_file(ctx, k ? PMP_NET_CHROMA_MSBD : PMP_NET_LUMA_MSBD, qp, k ? "Chroma_BD_22.pmpw" : "Luma_BD_22.pmpw"))) return rc;
        if ((rc = pmp_infer_postprocess(ctx, k ? PMP_CHROMA : PMP_LUMA, qp, by, bu, bv, n, hor, ver, qt, dire, 0, 0, 0))) return rc;
        if ((rc = pmp_tile_partition_maps(frames, h, w, hor, ver, qt, dire, H, V, Q, D))) return rc;
    }
    _Static_assert(PMP_LUMA == 0 && PMP_CHROMA == 1 && PMP_NET_CHROMA_Q == 2, "component ids are not net ids");
    _Static_assert(PMP_RECORD_BYTES == 256 + 256 + 64 + 768, "record layout");
    return pmp_set_saturation_policy(ctx, PMP_SAT_RERUN) | pmp_set_precision(ctx, PMP_PRECISION_F16X3);
}
'''


def test_documented_c_usage_compiles_as_plain_c(tmp_path):
    src = tmp_path / "snippet.c"
    src.write_text(SNIPPET)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])


def test_integration_md_uses_component_ids_for_inference_calls():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read() + open(os.path.join(ROOT, "tools", "vtm_build", "pmp_hook.cpp")).read()
    calls = re.findall(r"pmp_infer(?:_postprocess)?(?:_device)?\(\s*ctx,\s*([^,]+),", text)
    assert calls, "no documented inference call found"
    for arg in calls:
        assert "PMP_NET_" not in arg, "inference calls take PMP_LUMA / PMP_CHROMA, not a net id: " + arg
