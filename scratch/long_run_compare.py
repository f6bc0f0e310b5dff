"""usage: python scratch/long_run_compare.py a.npz b.npz  -- two scratch/long_run.py results (e.g. bf16 and fp16 + loss scale)"""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for r in (a, b):
    e = np.linalg.norm(r['means'] - r['rmeans'], axis=1) / np.linalg.norm(r['rmeans'], axis=1)
    s = np.abs(r['stds'].mean(1) - r['rstds'].mean(1))
    # nearest data class of every generated class mean: does the generator condition on the label?
    d = np.linalg.norm(r['means'][:, None, :] - r['rmeans'][None, :, :], axis=2)
    print(f"[{r['dtype']}] {int(r['iters'])} iterations in {float(r['seconds']):.0f} s, finite {bool(r['finite'])}")
    rms = np.sqrt(((r['means'] - r['rmeans']) ** 2).mean(1))
    sep = np.sqrt(((r['rmeans'][:, None, :] - r['rmeans'][None, :, :]) ** 2).mean(2))
    print(f"   class-mean error, RMS per pixel (tanh units): mean {rms.mean():.3f}  max {rms.max():.3f}  (data classes are {sep[sep > 0].min():.3f} .. {sep.max():.3f} apart);"
          f"  relative {e.mean():.3f};  labels recovered by nearest class mean: {(d.argmin(1) == np.arange(10)).sum()}/10")
    print(f"   per-pixel std, generated vs data (mean over pixels): {r['stds'].mean():.3f} vs {r['rstds'].mean():.3f}   (max class difference {s.max():.3f})")
    n = len(r['d_loss'])
    for lo, hi in ((0, n // 4), (n // 4, n // 2), (n // 2, 3 * n // 4), (3 * n // 4, n)):
        print(f"   iterations {25 * lo:5d}-{25 * hi:5d}: d_loss {r['d_loss'][lo:hi].mean():.3f}  g_loss {r['g_loss'][lo:hi].mean():.3f}")
d = np.linalg.norm(a['means'] - b['means'], axis=1) / np.linalg.norm(a['rmeans'], axis=1)
rms = np.sqrt(((a['means'] - b['means']) ** 2).mean(1))
print(f"between the two runs: class means RMS per pixel {rms.mean():.3f} (max {rms.max():.3f}); relative to |mean_data| {d.mean():.3f}")
