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
