"""The assembly interpreters call compiled leaf functions with s_swappc_b64 (TI_CALL / MPR_CALL /
NQ_CALLC).  Their asm statements declare as clobbered exactly the caller-saved registers those
functions may touch; the interpreters' own state sits in registers the AMDGPU calling convention
preserves (v40-v47, v56-v63, SGPRs from s34 up).  That holds only while the callees stay inside the
budget — this test recompiles the three files to assembly (hipcc cross-compiles without a GPU) and
reads the per-function register counts the compiler records."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpr_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# file, symbol prefix, VGPR budget (the clobber lists cover v0..v39, v48..v55, v64..v71 for the interval routines —
# but the walk with the slot file in registers keeps slot 1 in v70 / v71, so v69 is their last — and v0..v31 for the others), SGPR budget (s0..s31, the return address s[30:31] among them)
CASES = [("kernels.hip", "mpr_ti_", ["asin", "acos", "atan", "exp", "log", "divx", "sqrtx"], 70, 32),
         ("kernels_voxel_asm.hip", "mpr_fa_", ["asin", "acos", "atan"], 32, 32),
         ("kernels_normals_asm.hip", "mpr_nq_", ["asin", "acos", "atan"], 32, 32),
         # the generated-code float pass keeps its own values in v8..v31 / s16..s29: the routines get v0..v7, s0..s15
         ("kernels_voxel_jit.hip", "mpr_fj_", ["asin", "acos", "atan"], 8, 16)]


@pytest.mark.parametrize("src,prefix,names,vgprs,sgprs", CASES, ids=[c[0] for c in CASES])
def test_called_routines_stay_inside_the_clobber_lists(src, prefix, names, vgprs, sgprs):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "x.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-mllvm", "-structurizecfg-skip-uniform-regions=1", "-I" + CSRC,
                               "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                               os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read()
    for n in names:
        sym = prefix + n

        def field(name):
            m = re.search(r"\.set \.L%s\.%s, (\d+)" % (re.escape(sym), name), text)
            assert m, "%s: no %s record for %s" % (src, name, sym)
            return int(m.group(1))

        assert field("num_vgpr") <= vgprs, "%s uses v%d: beyond what the asm statement declares clobbered" % (sym, field("num_vgpr") - 1)
        body = text[text.index("\n%s:" % sym):]
        body = body[: body.index(".Lfunc_end")]
        if sgprs >= 32:
            assert field("numbered_sgpr") <= sgprs, "%s uses s%d" % (sym, field("numbered_sgpr") - 1)
        else:
            # the return address s[30:31] always counts; look at the registers the body really names
            used = {int(n) for n in re.findall(r"\bs(\d+)\b", body)}
            for lo, hi in re.findall(r"\bs\[(\d+):(\d+)\]", body):
                used.update(range(int(lo), int(hi) + 1))
            used -= {30, 31}
            assert max(used, default=0) < sgprs, "%s uses s%d" % (sym, max(used))
        # no stack: a leaf that needed callee-saved registers (v40-v47, v56-v63, s34+) would have to spill them
        assert field("private_seg_size") == 0, "%s uses the stack" % sym
        assert field("num_agpr") == 0
        # the interpreter passes the return address in s[30:31]
        assert "s_setpc_b64 s[30:31]" in body, "%s does not return through s[30:31]" % sym
        assert "s_swappc_b64" not in body, "%s is not a leaf" % sym
