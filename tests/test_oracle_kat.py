"""Known-answer tests for the first-party pieces of the path (reference src/main.rs) — SURVEY §8c."""
import numpy as np
import pytest

from oracle import oracle as orc


def test_mel_frame_count():
    # src/main.rs:444-452: 1 + n/160, minus one when > 1
    assert orc.mel_frames(480000) == 3000
    assert orc.mel_frames(160) == 1
    assert orc.mel_frames(1) == 1
    assert orc.mel_frames(159) == 1
    assert orc.mel_frames(320) == 2
    assert orc.mel_frames(16000 * 301 + 9184) == (16000 * 301 + 9184) // 160


def test_empty_audio_rejected():
    with pytest.raises(ValueError, match="Empty audio"):  # src/main.rs:414-416
        orc.log_mel(np.zeros(0, np.float32))


def test_filterbank_properties():
    fb = orc.mel_filterbank(80)
    assert fb.shape == (80, 201) and fb.dtype == np.float32
    assert (fb >= 0).all()
    assert fb[:, 0].sum() == 0.0  # DC bin is at f_left of filter 0
    # HF / librosa Slaney filterbank, built independently in float64
    def hz2mel(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-9) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)
    def mel2hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)
    pts = mel2hz(np.linspace(hz2mel(0.0), hz2mel(8000.0), 82))
    freqs = np.arange(201) * 8000.0 / 200.0
    ref = np.zeros((80, 201))
    for m in range(80):
        lo = (freqs - pts[m]) / (pts[m + 1] - pts[m])
        up = (pts[m + 2] - freqs) / (pts[m + 2] - pts[m + 1])
        ref[m] = np.maximum(0, np.minimum(lo, up)) * 2.0 / (pts[m + 2] - pts[m])
    np.testing.assert_allclose(fb, ref, rtol=0, atol=2e-6)  # f32 vs f64 build ≈1.2e-7


def test_mel_silence_and_short():
    # all-zero audio: every mel energy clamps to 1e-10 → log10 = -10 → (−10+4)/4 = −1.5
    mel = orc.log_mel(np.zeros(16000, np.float32))
    assert mel.shape == (80, 100)
    np.testing.assert_array_equal(mel, np.full((80, 100), -1.5, np.float32))
    # a single sample (len < 2 → zero padding branch, src/main.rs:432-435), one frame
    mel1 = orc.log_mel(np.array([0.5], np.float32))
    assert mel1.shape == (80, 1) and np.isfinite(mel1).all()
    # global max normalisation: values live in [(max-8+4)/4, (max+4)/4]
    x = np.sin(np.arange(8000) * 0.3).astype(np.float32)
    m = orc.log_mel(x)
    assert m.max() - m.min() <= 2.0 + 1e-6


def test_window_mel_zero_pads_in_normalised_space():
    # src/main.rs:895-905: a short tail is padded with 0.0 AFTER normalisation
    full = np.arange(80 * 10, dtype=np.float32).reshape(80, 10) + 1.0
    w = orc.window_mel(full, 4, 3000)
    assert w.shape == (80, 3000)
    np.testing.assert_array_equal(w[:, :6], full[:, 4:])
    assert (w[:, 6:] == 0).all()
    assert (orc.window_mel(full, 10, 3000) == 0).all()


def test_argmax_semantics():
    # src/main.rs:709-735
    row = np.array([[0.0, 5.0, 5.0, -1.0]], np.float32)
    assert orc.argmax_last_row(row) == 1                     # strict > : lowest index wins ties
    assert orc.argmax_last_row(row, suppress=[1]) == 2
    assert orc.argmax_last_row(row, suppress=[1, 2]) == 0
    nan = np.array([[np.nan, 1.0, np.nan]], np.float32)
    assert orc.argmax_last_row(nan) == 1                     # NaN never wins
    assert orc.argmax_last_row(np.array([[np.nan, np.nan]], np.float32)) == 0
    assert orc.argmax_last_row(np.full((1, 3), -np.inf, np.float32)) == 0
    two = np.array([[[9.0, 0.0, 0.0], [0.0, 0.0, 3.0]]], np.float32)
    assert orc.argmax_last_row(two) == 2                     # LAST row only
    with pytest.raises(ValueError):
        orc.argmax_last_row(np.zeros(4, np.float32))         # ndim < 2 → bail
