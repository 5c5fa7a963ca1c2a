"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol, its host-only
logic matches the reference semantics, and compute entry points fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from whisper_rust_ort_amd import binding as wb
from whisper_rust_ort_amd import modelspec as ms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = wb.load_library()
    hdr = open(os.path.join(ROOT, "include", "whisper_hip.h")).read()
    declared = set(re.findall(r"\b(wh_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(wb.EXPORTS), declared ^ set(wb.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), f"libwhisper_hip.so does not export {name}"
    assert lib.wh_abi_version() == 1


def test_mel_frames_matches_reference_rule():
    lib = wb.load_library()
    for n in (1, 2, 159, 160, 161, 319, 320, 480000, 4825184):
        expect = 1 + n // 160          # src/main.rs:444-448
        if expect > 1:
            expect -= 1                # :450-452
        assert lib.wh_mel_frames(n) == expect


def test_longform_plan_matches_reference_loop():
    def ref(n, chunk_s=30.0, ov_s=5.0):  # src/main.rs:858-861, 875-882
        chunk_len = int(round(np.float32(chunk_s) * np.float32(16000)))
        overlap = int(round(np.float32(ov_s) * np.float32(16000)))
        step = max(1, chunk_len - overlap if chunk_len > overlap else 0)
        out, pos = [], 0
        while pos < n:
            end = min(pos + chunk_len, n)
            out.append(pos)
            if end == n:
                break
            pos += step
        return out
    for n in (1, 479999, 480000, 480001, 880000, 880001, 4825184):
        assert wb.longform_plan(n) == ref(n)
    assert wb.longform_plan(1000000, 10.0, 2.5) == ref(1000000, 10.0, 2.5)
    assert wb.longform_plan(480000) == [0]        # a 30 s clip is exactly one window
    assert len(wb.longform_plan(4825184)) == 12   # the 301.574 s file → 12 windows
    assert wb.longform_plan(0) == []


@pytest.mark.parametrize("preset,seed", [("nano", 7), ("micro", 11)])
def test_cpp_synthetic_generator_is_bit_identical_to_numpy(preset, seed):
    a = wb.synthetic_weights(preset, seed)
    b = ms.flatten_state_dict(ms.PRESETS[preset], ms.synth_state_dict(ms.PRESETS[preset], seed))
    assert a.dtype == b.dtype == np.float32 and a.size == b.size == ms.n_params(ms.PRESETS[preset])
    np.testing.assert_array_equal(a, b)


def test_compute_fails_loudly_without_gpu():
    if wb.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(wb.WhisperHipError) as ei:
        wb.Model("synthetic:nano:7")
    assert ei.value.code == 6 and "no CPU fallback" in str(ei.value)


def test_bad_specs_are_reported():
    with pytest.raises(wb.WhisperHipError) as ei:
        wb.Model("synthetic:not-a-preset:1")
    assert ei.value.code == 4
    with pytest.raises(wb.WhisperHipError) as ei:
        wb.Model("/nonexistent/model/dir")
    assert ei.value.code == 7 and "config.json" in str(ei.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "whisper-rust-ort_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "libwhisper_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn


def test_cu_mask_helper_and_ctx_opts_validation():
    """wh_ctx_create_ex validates its options before it touches HIP (so this runs without a GPU): struct size, NULL masks, empty masks,
    masks that leave an XCD unrestricted (bit i = compute unit i / 8 of XCD i % 8: an XCD without a bit is not restricted at all)."""
    w = wb.cu_mask(0, 64)
    assert w.dtype == np.uint32 and w.size == 8 and int(w[0]) == 0xFFFFFFFF and int(w[1]) == 0xFFFFFFFF and not w[2:].any()
    assert sum(bin(int(x)).count("1") for x in wb.cu_mask(64, 192)) == 192
    with pytest.raises(ValueError):
        wb.cu_mask(200, 100)
    lib = wb.load_library()
    fake_model = C.create_string_buffer(1 << 16)          # never dereferenced: every case below fails validation first
    u32p = C.POINTER(C.c_uint32)

    def create(struct_size, max_batch, enc, dec, enc_words=None, dec_words=None):
        e = np.ascontiguousarray(enc, np.uint32) if enc is not None else None
        d = np.ascontiguousarray(dec, np.uint32) if dec is not None else None
        o = wb.WhCtxOpts(struct_size, max_batch, 0,
                         e.ctypes.data_as(u32p) if e is not None else None, (e.size if e is not None else 0) if enc_words is None else enc_words,
                         d.ctypes.data_as(u32p) if d is not None else None, (d.size if d is not None else 0) if dec_words is None else dec_words)
        h = C.c_void_p()
        rc = lib.wh_ctx_create_ex(C.cast(fake_model, C.c_void_p), C.byref(o), C.byref(h))
        return rc, (lib.wh_last_error(None) or b"").decode()

    ok_size = C.sizeof(wb.WhCtxOpts)
    rc, msg = create(ok_size - 8, 4, None, None)
    assert rc == 4 and "struct_size" in msg
    rc, msg = create(ok_size, 0, None, None)
    assert rc == 4 and "max_batch" in msg
    rc, msg = create(ok_size, 4, None, None, enc_words=8)
    assert rc == 4 and "NULL CU mask" in msg
    rc, msg = create(ok_size, 4, np.zeros(8, np.uint32), None)
    assert rc == 4 and "no bit set" in msg
    one_xcd = np.zeros(8, np.uint32)
    one_xcd[0] = 0x01010101                                # compute units 0..3 of XCD 0 only
    rc, msg = create(ok_size, 4, one_xcd, None)
    assert rc == 4 and "every XCD" in msg
    rc, msg = create(ok_size, 4, wb.cu_mask(0, 64), wb.cu_mask(3, 5))   # bits 3..7: XCDs 0..2 uncovered
    assert rc == 4 and "every XCD" in msg
