import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import oracle as orc
from whisper_rust_ort_amd import binding as wb, modelspec as ms
preset, seed, clip = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dims = ms.PRESETS[preset]
sd = ms.synth_state_dict(dims, seed)
wq = ms.flatten_state_dict(dims, ms.fake_quant_state_dict(sd))
pcm = ms.synth_clip(clip)
mel = orc.window_mel(orc.log_mel(pcm, dims.n_mels), 0, 3000)
e_mx = orc.encoder(dims, wq, mel, act_mx=True)
e_no = orc.encoder(dims, wq, mel, act_mx=False)
model = wb.Model(f"synthetic:{preset}:{seed}", 0, wb.WH_PREC_FP8)
ctx = wb.Context(model, 1)
e_hip = ctx.run_encoder(ctx.whisper_log_mel(pcm))
def st(a, b): d = np.abs(a - b); return f"mean {d.mean():.4f} max {d.max():.4f}"
print("enc scale: std", e_no.std())
print("oracle mx vs oracle no-mx:", st(e_mx, e_no))
print("hip vs oracle mx       :", st(e_hip, e_mx))
print("hip vs oracle no-mx    :", st(e_hip, e_no))
