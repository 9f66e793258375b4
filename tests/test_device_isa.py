"""The device code of libv2e_amd.so holds no packed-float32 VALU instruction.

Round 4 (scripts/concurrency_repro.py): built WITH the compiler's SLP vectoriser (which turns adjacent scalar
float32 operations into v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32), a UNet pass running beside bf16 / f16 MFMA kernels of another
stream came out wrong in 10 runs of 10 on the MI355X; built with -fno-slp-vectorize (v2e_amd/csrc/Makefile) never.  Round 5 found the
mechanism (profiles/r05_concurrency_rootcause.txt, scripts/pk_opsel_mfma_repro.hip): a packed-float32 instruction whose op_sel routes
the HIGH half of src1 to the LOW lane reads 0.0 there while another kernel issues independent 16-K bf16 / f16 MFMAs -- a hardware
interaction, which only the vectoriser's shuffled pairs ever produced here.  This test disassembles what was built, so that neither
the flag nor the rule can be lost silently; on a host with a GPU (where the hazard exists) missing tools are a failure, not a skip."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _device_disassembly(tmp_path):
    so = os.path.join(ROOT, "v2e_amd", "csrc", "libv2e_amd.so")
    objcopy, bundler, objdump = (os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"))
    if not (os.path.isfile(so) and all(os.path.isfile(t) for t in (objcopy, bundler, objdump))):
        from conftest import has_gpu
        if has_gpu():
            pytest.fail("libv2e_amd.so or the LLVM binary tools (%s) are missing on a GPU host: the packed-float32 rule is unchecked" % LLVM)
        pytest.skip("libv2e_amd.so or the LLVM binary tools are not here")
    fat = str(tmp_path / "fat.bin")
    subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, "no offload bundle in .hip_fatbin"
    text = []
    for i, a in enumerate(starts):  # one bundle per translation unit
        part = str(tmp_path / ("bundle%d.bin" % i))
        with open(part, "wb") as f:
            f.write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        elf = str(tmp_path / ("co%d.elf" % i))
        subprocess.run([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part,
                        "--output=" + elf], check=True)
        text.append(subprocess.run([objdump, "-d", elf], check=True, capture_output=True, text=True).stdout)
    return "\n".join(text)


def test_no_packed_float32_instructions_in_the_device_code(tmp_path):
    dis = _device_disassembly(tmp_path)
    assert len(re.findall(r"\bv_mfma_f32_32x32x16_(bf16|f16)\b", dis)) > 100 and "k_chain" in dis  # it IS the library's device code
    # the form that is actually hazardous: op_sel with src1's high half routed to the low lane (op_sel:[x,1])
    hazard = re.findall(r"\bv_pk_(?:mul|add|fma)_f32\b[^\n]*op_sel:\[[01],1", dis)
    assert not hazard, "%d packed-float32 instructions with op_sel:[.,1] (src1.hi -> low lane): wrong beside 16-K MFMAs on gfx950" % len(hazard)
    # and, as the build rule, none at all (the vectoriser decides the operand routing, not the source)
    packed = re.findall(r"\bv_pk_(?:mul|add|fma)_f32\b", dis)
    assert not packed, "%d packed-float32 instructions: built without -fno-slp-vectorize?" % len(packed)
