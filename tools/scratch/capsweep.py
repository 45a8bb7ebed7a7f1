import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import capture_rate as cr
cfg1 = dict(Fs=240000, Rs=10000, M=2, P=24, f1=10000, shift=10000, est_min=500, est_max=25000)
slots = int(os.environ.get("SLOTS", "4096"))
for eb in (9.0, 8.0, 7.0, 6.0, 5.0, 4.0):
    d = cr.synth(cfg1, 100_000_000, eb, 5)
    cr.run("2-FSK -p 24, Eb/N0 %g dB" % eb, cfg1, d, slots)
d = cr.synth(cfg1, 24_000_000, 8.0, 7)
cr.run("8 dB +30 ppm", cfg1, cr.resample_host(d, 30e-6), slots)
cr.run("8 dB -100 ppm", cfg1, cr.resample_host(d, -100e-6), slots)
