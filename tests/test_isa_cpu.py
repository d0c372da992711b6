"""Guards on the machine code of the GEMM kernels, checked on the CPU by disassembling the gfx950 code object inside
the built libwesup_hip.so (no GPU needed).

The kernels stage their operands with LDS-DMA through inline asm that writes M0 and leaves it (csrc/gemm.hip, "M0
contract"); hipcc is told so through the clobber list but M0 is a register it reserves for itself, so the contract
is enforced here instead of trusted:
  * no GEMM kernel reads M0 except through the LDS-DMA instructions (and the save/restore of the image layer's 64-bit
    form): no s_movrel / v_movrel, no GWS / add-tid DS instruction, no s_sendmsg;
  * no scratch (spill) instruction in any GEMM kernel;
  * the unrolled K-step of every instantiation holds exactly the MFMAs it should: BK/2 x WM x WN
    v_mfma_f32_32x32x2_f32 -- a lost unroll (seen once: a run-time branch in the border arithmetic stopped hipcc from
    unrolling) or a duplicated loop shows up as another count.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
BK = 32


@pytest.fixture(scope='module')
def kernels(tmp_path_factory):
    """{mangled kernel name: [instruction lines]} of every code object in the library."""
    from wesup_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    if not os.path.exists(OBJDUMP):
        pytest.skip('llvm-objdump not found')
    work = tmp_path_factory.mktemp('isa')
    so = shutil.copy(_lib.LIB_PATH, work / 'lib.so')          # --offloading writes the bundles next to its input
    subprocess.run([OBJDUMP, '--offloading', str(so)], cwd=work, check=True, capture_output=True)
    out = {}
    for co in sorted(work.glob('lib.so.*gfx950')):
        text = subprocess.run([OBJDUMP, '-d', str(co)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in text.splitlines():
            m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
            if m:
                name = m.group(1)
                out[name] = []
            elif name and line.startswith('\t'):
                out[name].append(line.strip().split('//')[0].strip())
    assert out, 'no gfx950 code object found in the library'
    return out


def _gemm(kernels):
    return {k: v for k, v in kernels.items() if 'gemm_nt_kernel' in k or 'gemm_tn_kernel' in k}


def _dma_kernels(kernels):
    """Every kernel that stages through the LDS-DMA inline asm (the M0 contract applies to all of them): the GEMM family
    and the fused Winograd products + output transform (csrc/wino_fused.hip)."""
    return {k: v for k, v in kernels.items() if 'gemm_nt_kernel' in k or 'gemm_tn_kernel' in k or 'wino4_gemm_out_kernel' in k}


def test_every_gemm_instantiation_is_there(kernels):
    names = list(_gemm(kernels))
    assert sum('gemm_nt_kernel' in n for n in names) >= 24 and sum('gemm_tn_kernel' in n for n in names) >= 20, names


def test_no_scratch_and_no_foreign_m0_reader(kernels):
    dma = re.compile(r'^(buffer_load_dword\w* .* lds|global_load_lds_dword\w*)')
    assert any('wino4_gemm_out_kernel' in k for k in kernels)
    for name, code in _dma_kernels(kernels).items():
        n_dma = 0
        for ins in code:
            op = ins.split()[0]
            assert not op.startswith('scratch_'), (name, ins)
            assert 'movrel' not in op and not op.startswith('s_sendmsg') and 'gws' not in op and 'addtid' not in op, (name, ins)
            if dma.match(ins):
                n_dma += 1
            elif re.search(r'\bm0\b', ins):
                # the only explicit uses: writing M0 in front of a DMA, saving / restoring it around the 64-bit form
                assert re.match(r'^s_mov_b32 (m0, \S+|s\d+, m0)$', ins), (name, ins)
        assert n_dma >= 4, (name, n_dma)                       # and the staging really is LDS-DMA


def test_no_clock_read_in_the_shipped_gemm_kernels(kernels):
    """The in-kernel clock probe and the per-block trace are debug instrumentation (csrc/gemm.hip, WESUP_GEMM_DEBUG): they
    exist only in `make debug`'s libwesup_hip_debug.so; the shipped kernels read no clock and no trace pointer."""
    if os.environ.get('WESUP_HIP_LIB'):
        pytest.skip('another build of the library was selected through WESUP_HIP_LIB')
    for name, code in _gemm(kernels).items():
        for ins in code:
            op = ins.split()[0]
            assert op not in ('s_memrealtime', 's_memtime') and not op.startswith('s_getreg'), (name, ins)


def test_mfma_count_of_the_unrolled_k_step(kernels):
    for name, code in _gemm(kernels).items():
        n = sum(ins.startswith('v_mfma_f32_32x32x2_f32') or ins.startswith('v_mfma_f32_32x32x2f32') for ins in code)
        other = sum(ins.startswith('v_mfma') for ins in code) - n
        args = [int(a) for a in re.findall(r'Li(\d+)E', name)]
        extra = 0
        if 'gemm_nt_kernel' in name:          # <NW, BM, BN, WM, WN, MODE, MINB, RELU, SIDE>
            wm, wn = args[3], args[4]
            if re.findall(r'Lb([01])E', name)[1] == '1':
                # the fused side conv: K = BN in 2-wide MFMA steps, once per epilogue slab of the output tile
                bm, bn = args[1], args[2]
                lds_floats = 2 * (bm + bn) * BK
                slabs = -(-bm * (bn + 4) // lds_floats)
                extra = slabs * bn // 2
        else:                                 # <BM, BN, WM, WN, MODE, RELU, TINY>
            wm, wn = args[2], args[3]
        assert other == 0 and n == BK // 2 * wm * wn + extra, (name, n, other, wm, wn, extra)


FP32_MFMA = ('v_mfma_f32_32x32x2_f32', 'v_mfma_f32_32x32x2f32', 'v_mfma_f32_16x16x4_f32', 'v_mfma_f32_16x16x4f32')


def test_fp32_mfma_only(kernels):
    """north_star: fp32 arithmetic -- no reduced-precision matrix instruction anywhere in the library (the GEMM family uses
    the 32x32x2 fp32 MFMA, the fused Winograd kernel the 16x16x4 one: both multiply and accumulate in fp32)."""
    for name, code in kernels.items():
        for ins in code:
            if ins.startswith('v_mfma') or ins.startswith('v_smfmac'):
                assert ins.startswith(FP32_MFMA), (name, ins)


def test_fused_winograd_kernel_holds_its_mfmas(kernels):
    """wino4_gemm_out_kernel<KC>: six positions are unrolled per pass of the position-row loop, 32 KC MFMAs each (K = 64 KC
    in steps of 4, two 16x16 blocks per wave); a lost unroll shows up as another count, a spill as a scratch instruction."""
    for name, code in kernels.items():
        if 'wino4_gemm_out_kernel' not in name:
            continue
        kc = int(re.findall(r'Li(\d+)E', name)[0])
        n = sum(ins.startswith(('v_mfma_f32_16x16x4_f32', 'v_mfma_f32_16x16x4f32')) for ins in code)
        assert n in (6 * 32 * kc, 36 * 32 * kc), (name, n)         # row loop rolled or fully unrolled
        assert not any(ins.split()[0].startswith('scratch_') for ins in code), name
