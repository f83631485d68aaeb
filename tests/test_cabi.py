"""C-ABI checks that need no GPU: the shared library loads, exports every symbol that
include/beso_hip.h declares, and rejects bad configs / shapes before touching the device."""
import ctypes as C
import os
import re

import pytest

from beso_amd import _lib
from beso_amd.runtime import ScoreNetShape

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from beso_amd.build import build
    build(verbose=False)
    return _lib.load()


def test_library_is_built_from_the_sources_in_the_tree(lib):
    """build() keys on a content hash of csrc/ + the header (not on modification times): after it, the shipped .so is
    the one these sources produce; touching a source makes it stale."""
    from beso_amd import build as B
    assert B.is_current()
    src = os.path.join(B.CSRC, "feed.hip")
    text = open(src).read()
    try:
        open(src, "a").write("\n// probe\n")
        assert not B.is_current()
    finally:
        open(src, "w").write(text)
    assert B.is_current()


def declared_symbols(header="beso_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(beso_[a-z_0-9]+)\s*\(", text)))


def exported_symbols(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("beso_")})


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 12
    assert sorted(_lib.EXPORTS) == names, "binding list and header disagree"
    for n in names:
        assert getattr(lib, n) is not None


def test_product_library_has_no_development_entry_points(lib):
    """libbeso_hip.so exports exactly what include/beso_hip.h declares: no beso_debug_* symbol, nothing that mutates
    process-wide state (what a call may vary travels in its flags; the launch-site timers are thread-local).  The
    development aids of include/beso_hip_debug.h exist in libbeso_hip_dev.so only."""
    from beso_amd.build import build
    assert exported_symbols(_lib.LIB_PATH) == declared_symbols()
    assert not [n for n in exported_symbols(_lib.LIB_PATH) if "debug" in n]
    dev = build(verbose=False, dev=True)
    assert dev == _lib.DEV_LIB_PATH
    assert exported_symbols(dev) == sorted(declared_symbols() + declared_symbols("beso_hip_debug.h"))
    assert sorted(_lib.DEV_EXPORTS) == declared_symbols("beso_hip_debug.h")


def test_no_unpadded_mixed_shape_mfma_chains(lib):
    """The MI355X matrix-pipe hazard of DESIGN.md 4.1c, checked on the ISA of the shipped libraries (llvm-objdump, no
    GPU): no MFMA takes as SrcC the result of an MFMA of another shape fewer than ten wait states after it -- LLVM
    pads nothing there, and a half k-step issued two instructions behind the last full one made one instance of the
    kernel nondeterministic in round 3 (tools/check_mfma_chains.py; fused.hip mixed_chain_pad)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_mfma_chains as chk
    if not os.path.exists(os.path.join(chk.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    libs = [_lib.LIB_PATH] + ([_lib.DEV_LIB_PATH] if os.path.exists(_lib.DEV_LIB_PATH) else [])
    for path in libs:
        bad, n_fn, n_mfma = chk.check_library(path)
        assert n_fn > 50 and n_mfma > 10000, (path, n_fn, n_mfma)          # the disassembly was actually read
        assert not bad, (path, bad[:5])


def test_version_and_status_strings(lib):
    assert b"gfx950" in lib.beso_version()
    assert lib.beso_status_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -6):
        assert lib.beso_status_string(code) not in (b"ok", b"unknown status")


def test_sizes_for_shipped_shapes(lib):
    kitchen = ScoreNetShape(30, 9, 360, 6, 6, 2, 4, True, 0.5).c_struct()
    assert lib.beso_num_params(C.byref(kitchen)) == 107           # 113 state_dict entries - 6 mask buffers
    bf16 = lib.beso_packed_bytes(C.byref(kitchen), _lib.PREC_BF16)
    fp32 = lib.beso_packed_bytes(C.byref(kitchen), _lib.PREC_FP32)
    # bf16 image = generic GEMM operands (bf16) + the fused kernels' fragment-ordered copy
    assert 18_000_000 < bf16 < 90_000_000 and 37_000_000 < fp32 < 90_000_000
    ws1 = lib.beso_workspace_bytes(C.byref(kitchen), 4096, 4, _lib.PREC_BF16, 0)
    ws2 = lib.beso_workspace_bytes(C.byref(kitchen), 4096, 4, _lib.PREC_BF16, 1)
    assert 0 < ws1 < ws2 < 4 * ws1
    # t outside [1, W] is a shape error (score_gpts.py:282), reported as size 0 here
    assert lib.beso_workspace_bytes(C.byref(kitchen), 4, 5, _lib.PREC_BF16, 0) == 0
    assert lib.beso_workspace_bytes(C.byref(kitchen), 0, 4, _lib.PREC_BF16, 0) == 0


def test_bad_config_is_rejected(lib):
    bad = ScoreNetShape(30, 9, 361, 6, 6, 2, 4, True, 0.5).c_struct()    # D % H != 0
    assert lib.beso_num_params(C.byref(bad)) == 0
    assert lib.beso_packed_bytes(C.byref(bad), _lib.PREC_BF16) == 0
    st = lib.beso_denoise_fwd(C.byref(bad), None, 0, None, None, None, None, None, 1, 1, 0, 1.0, None, 0, None)
    assert st == -1
    with pytest.raises(ValueError):
        _lib.check(st, "denoise")


def test_null_and_shape_errors_do_not_touch_the_device(lib):
    cfg = ScoreNetShape(7, 3, 48, 2, 6, 2, 3, True, 0.5).c_struct()
    one = C.c_void_p(16)
    # t > obs_seq_len
    st = lib.beso_denoise_fwd(C.byref(cfg), one, 0, one, one, one, one, one, 2, 9, 0, 1.0, one, 1 << 30, None)
    assert st == -2
    # null pointers
    st = lib.beso_denoise_fwd(C.byref(cfg), None, 0, one, one, one, one, one, 2, 2, 0, 1.0, one, 1 << 30, None)
    assert st == -3
    # unknown precision / flag / sampler
    assert lib.beso_denoise_fwd(C.byref(cfg), one, 7, one, one, one, one, one, 2, 2, 0, 1.0, one, 1 << 30, None) == -3
    assert lib.beso_denoise_fwd(C.byref(cfg), one, 0, one, one, one, one, one, 2, 2, 8, 1.0, one, 1 << 30, None) == -3
    sig = (C.c_float * 3)(1.0, 0.5, 0.0)
    assert lib.beso_sample(C.byref(cfg), one, 0, 9, one, one, one, 2, 2, sig, 3, 1.0, 0, one, 1 << 30, None) == -3
    assert lib.beso_sample(C.byref(cfg), one, 0, 0, one, one, one, 2, 2, sig, 3, 1.0, 8, one, 1 << 30, None) == -3     # unknown flag
    # workspace too small
    st = lib.beso_denoise_fwd(C.byref(cfg), one, 0, one, one, one, one, one, 2, 2, 0, 1.0, one, 16, None)
    assert st == -4
    with pytest.raises(_lib.BesoHipError):
        _lib.check(st)
    # sampler step argument checks
    assert lib.beso_sampler_step(9, one, None, one, None, one, 1.0, 1.0, 4, None) == -3
    assert lib.beso_sampler_step(_lib.STEP_HEUN_CORRECT, one, one, one, None, one, 1.0, 1.0, 4, None) == -3
    assert lib.beso_sampler_step(_lib.STEP_DDIM, one, None, one, None, one, 1.0, 1.0, 0, None) == 0


def test_feed_argument_errors_do_not_touch_the_device(lib):
    """beso_gather_windows rejects NULLs, a window longer than the trajectories, an unknown goal mode and a random goal
    mode without draws before anything is enqueued (status -3), and an empty batch is a no-op."""
    one = C.c_void_p(0x1000)
    call = lambda **k: lib.beso_gather_windows(*[k.get(n, d) for n, d in (          # noqa: E731
        ("observations", one), ("actions", one), ("seq_len", one), ("n_traj", 4), ("t_max", 20), ("obs", 5), ("act", 3),
        ("slice_traj", one), ("slice_start", one), ("n_slices", 30), ("batch_slices", one), ("draws", one), ("batch", 8),
        ("window", 6), ("goal_len", 2), ("goal_mode", _lib.GOAL_RANDOM), ("sep", 0), ("obs_out", one), ("act_out", one),
        ("goal_out", one), ("stream", None))])
    assert call(batch=0) == 0
    for bad in (dict(observations=None), dict(batch_slices=None), dict(obs_out=None), dict(window=21), dict(window=0),
                dict(goal_mode=3), dict(draws=None), dict(goal_out=None), dict(goal_len=-1), dict(n_slices=0), dict(batch=-1)):
        assert call(**bad) == -3, bad
    assert call(batch=0, goal_mode=_lib.GOAL_TAIL, draws=None) == 0


def test_early_gradient_range_matches_parameter_layout(lib):
    """beso_grad_early_range = [offset of blocks[l0].ln1.weight, end of ln_f.bias) in the order of the parameters, 1 <= l0 < L."""
    import torch
    from beso_amd.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT
    for kw in (dict(state_dim=7, action_dim=3, embed_dim=48, n_layers=4, n_heads=6, goal_seq_len=2, obs_seq_len=3, linear_output=True),
               dict(state_dim=5, action_dim=2, embed_dim=32, n_layers=3, n_heads=4, goal_seq_len=1, obs_seq_len=4,
                    linear_output=False),
               dict(state_dim=5, action_dim=2, embed_dim=32, n_layers=1, n_heads=4, goal_seq_len=1, obs_seq_len=4, linear_output=True)):
        net = DiffusionGPT(device="cpu", goal_conditioned=True, embed_pdrob=0.0, attn_pdrop=0.0, resid_pdrop=0.0, **kw)
        cfg = net.shape(0.5).c_struct()
        b, e = C.c_size_t(7), C.c_size_t(7)
        assert lib.beso_grad_early_range(C.byref(cfg), C.byref(b), C.byref(e)) == 0
        offs, off = {}, 0
        for name, prm in net.named_parameters():
            offs[name] = (off, off + prm.numel())
            off += prm.numel()
        assert off == lib.beso_grad_floats(C.byref(cfg))
        L = kw["n_layers"]
        if L < 2:
            assert b.value == e.value
            continue
        starts = {offs[next(n for n in offs if n.startswith(f"blocks.{l}."))][0]: l for l in range(L)}
        assert b.value in starts and 1 <= starts[b.value] <= L - 1, (b.value, starts)     # the first parameter of an upper layer
        assert e.value == offs["ln_f.bias"][1]
    assert lib.beso_grad_early_range(None, C.byref(b), C.byref(e)) == -3


def test_product_path_has_no_cpu_fallback():
    """CPU tensors must raise, not silently compute somewhere else."""
    import torch
    from beso_amd.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT
    from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser
    inner = DiffusionGPT(state_dim=7, device="cpu", goal_conditioned=True, action_dim=3, embed_dim=48,
                         embed_pdrob=0, attn_pdrop=0, resid_pdrop=0, n_layers=2, n_heads=6, goal_seq_len=2,
                         obs_seq_len=3, sigma_vocab_size=3, time_embedding_fn=None, linear_output=True)
    m = GCDenoiser(inner, sigma_data=0.5).eval()
    s, a, g = torch.zeros(2, 3, 7), torch.zeros(2, 3, 3), torch.zeros(2, 2, 7)
    with torch.no_grad(), pytest.raises(RuntimeError, match="GPU"):
        m(s, a, g, torch.ones(2))
