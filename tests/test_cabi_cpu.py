"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/mmhip.h declares, and its host-side
logic (parameter layout, capacity checks) behaves -- no GPU work is enqueued here."""
import ctypes as C
import os
import re

import pytest

import smtc_amd  # noqa: F401
from smtc_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.lib()


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "mmhip.h")).read()
    declared = set(re.findall(r"\b(mmhip_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mmhip_engine"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def _cfg(**kw):
    base = dict(hidden=768, heads=12, inter=3072, layers_txt=2, layers_img=2, vocab=1000, max_pos=130, type_vocab=1, txt_kind=1,
                pad_id=1, ln_eps_txt=1e-5, ln_eps_img=1e-12, image=224, patch=16, proj_dim=512, num_labels=3, fusion=1,
                p_hidden=0.1, p_attn=0.1, p_head=0.05, dtype=0, max_posts=4, max_text_len=64, loss_scale=0.0)
    base.update(kw)
    return _lib.Config(**base)


def test_layout_matches_reference_state_dict(lib):
    from oracle import mm_oracle as O
    cfg = _cfg()
    h = C.c_void_p()
    assert lib.mmhip_create(C.byref(cfg), C.byref(h)) == 0
    ocfg = O.OracleConfig(layers_txt=2, layers_img=2, vocab=1000, max_pos=130, num_labels=3)
    want = O.param_shapes(ocfg)
    got, spans = {}, {0: [], 1: []}
    pi = _lib.ParamInfo()
    for i in range(lib.mmhip_param_count(h)):
        assert lib.mmhip_param_info_at(h, i, C.byref(pi)) == 0
        name = pi.name.decode()
        got[name] = tuple(pi.dims[: pi.ndim])
        spans[pi.buffer].append((pi.offset, pi.offset + pi.numel))
        assert pi.offset % 4 == 0
        assert (pi.buffer == 0) == (not O.trainable(name)), name      # frozen <=> 'vision' in a dual_encoder name
    assert got == {k: tuple(v) for k, v in want.items()}
    for b in (0, 1):                                                   # no overlap inside a flat buffer
        s = sorted(spans[b])
        assert all(s[i][1] <= s[i + 1][0] for i in range(len(s) - 1))
        assert s[-1][1] <= lib.mmhip_buffer_numel(h, b)
    # q/k/v weights of a layer are contiguous (packed QKV GEMM operand)
    idx = {n: i for i, n in enumerate(got)}
    p = "dual_encoder.text_model.encoder.layer.1.attention.self."
    offs = []
    for n in ("query", "key", "value"):
        lib.mmhip_param_info_at(h, idx[p + n + ".weight"], C.byref(pi))
        offs.append(pi.offset)
    assert offs[1] - offs[0] == 768 * 768 and offs[2] - offs[1] == 768 * 768
    assert lib.mmhip_num_backward_stages(h) == 4
    b, e = C.c_uint64(), C.c_uint64()
    prev_end = None
    for st in range(4):
        assert lib.mmhip_stage_grad_range(h, st, C.byref(b), C.byref(e)) == 0 and b.value < e.value
        if prev_end is not None:
            assert b.value == prev_end                                 # stages tile the active range in backward order
        prev_end = e.value
    assert prev_end == lib.mmhip_buffer_numel(h, 1)
    lib.mmhip_destroy(h)


def test_create_rejects_bad_configs(lib):
    h = C.c_void_p()
    for bad in (dict(hidden=700), dict(heads=8), dict(max_text_len=256), dict(dtype=3), dict(max_pos=64), dict(patch=15), dict(hidden_img=1000), dict(fusion=1, hidden_img=1024, heads_img=16)):
        cfg = _cfg(**bad)
        assert lib.mmhip_create(C.byref(cfg), C.byref(h)) == -1, bad
    assert lib.mmhip_forward(None, None, None, None, None, None, 1, 1, 0, 0, None, None, None, None, None) == -2
