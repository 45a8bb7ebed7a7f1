import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import capture_rate as cr
cfg1 = dict(Fs=240000, Rs=10000, M=2, P=24, f1=10000, shift=10000, est_min=500, est_max=25000)
d = cr.synth(cfg1, 100_000_000, 5.0, 5)
cr.run("5 dB", cfg1, d, 4096)
