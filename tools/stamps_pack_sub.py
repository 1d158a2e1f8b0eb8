#!/usr/bin/env python3
"""Phase timeline of k_pack_sub workgroups (a workgroup per written subframe) from in-kernel time stamps.
Needs a build of enc_pack.hip with -DPS_STAMPS (tools/build_variant.sh psstamps enc_pack.hip -DPS_STAMPS) selected through
FLAC_AMD_LIB; the stamps are the 100 MHz wall clock scaled to shader cycles, so that they compare across the chip's dies.
usage: FLAC_AMD_LIB=$PWD/flac_amd/lib/exp/psstamps.so python tools/stamps_pack_sub.py [C4|C3]"""
import os, sys
os.environ["FLAC_AMD_STAMPS"] = "pack"
sys.path.insert(0, ".")
import numpy as np, torch
import flac_amd
from flac_amd import workloads as W
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
ch, bps, rate, secs = (8, 24, 192000, 60) if cfg == "C4" else (2, 24, 96000, 150)
from flac_amd.synth import synth_pcm
pcm = synth_pcm(seconds=secs, seed=1234, sample_rate=rate, channels=ch, bits_per_sample=bps).astype(np.int32)
over = {} if cfg == "C4" else {"do_exhaustive_model_search": 1}
p = flac_amd.preset(8, ch, bps, rate, 4096, **over)
nfr = (pcm.shape[0] + 4095) // 4096
dev = torch.device("cuda", 0)
pcm_dev = torch.from_numpy(pcm).to(dev)
enc = flac_amd.Encoder(p, 0, nfr)
cap = enc.max_output_bytes(nfr)
out = torch.empty(cap, dtype=torch.uint8, device=dev)
fb = torch.zeros(nfr, dtype=torch.int32, device=dev); offs = torch.zeros(nfr + 1, dtype=torch.int64, device=dev)
for _ in range(3):
    enc.encode_device(pcm_dev.data_ptr(), 4, pcm.shape[0], out.data_ptr(), cap, fb.data_ptr(), offs.data_ptr(), first_frame=0, is_last=True, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
n = nfr * ch
raw = enc.debug_read(1, n * 128).view(np.uint64).reshape(n, 16)
t = raw[:-ch, :13].astype(np.int64)
names = ["loads issued+arrived", "stage store+sync", "values", "scan sync", "size out, gather, attempt, zero", "sync", "emit (+attempt)", "sync", "crc", "tail/crc exchange, place", "sync", "flush"]
d = np.diff(t, axis=1)
ok = (d >= 0).all(axis=1) & (d < 10_000_000).all(axis=1)
print("pieces %d (%d usable); cycles per phase: median / mean / p90" % (n, ok.sum()))
for i, nm in enumerate(names):
    print("  %-34s %8.0f %8.0f %8.0f" % (nm, np.median(d[ok, i]), d[ok, i].mean(), np.percentile(d[ok, i], 90)))
tot = t[ok][:, 12] - t[ok][:, 0]
print("  %-34s %8.0f %8.0f %8.0f" % ("total", np.median(tot), tot.mean(), np.percentile(tot, 90)))
full = raw[:-ch].astype(np.int64)
pl = full[:, 14] - full[:, 13]
print("  wave 1: blocking place %8.0f %8.0f %8.0f   already placed %.2f   enters at (since start) %8.0f" % (np.median(pl[ok]), pl[ok].mean(), np.percentile(pl[ok], 90), (full[ok, 15] & 1).mean(), np.median(full[ok, 13] - full[ok, 0])))
it = (full[ok, 15] >> 8) & 0xffff; sl = (full[ok, 15] >> 24) & 0xffffff; nn = (full[ok, 15] >> 48) & 0xff
print("  look-back passes: median %d mean %.1f p90 %d; of them retried after a sleep: median %d mean %.1f p90 %d; tickets missing at the last retry: mean %.1f" % (np.median(it), it.mean(), np.percentile(it, 90), np.median(sl), sl.mean(), np.percentile(sl, 90), nn.mean()))
# how far apart the pieces of a frame start, and channel 7 against channel 0
st = full[:, 0].reshape(-1, ch)
print("  start of piece c minus start of piece 0 (median):", [int(np.median(st[:, c] - st[:, 0])) for c in range(ch)])
en = full[:, 12].reshape(-1, ch)
print("  end of piece c minus start of piece 0 (median):", [int(np.median(en[:, c] - st[:, 0])) for c in range(ch)])
print("  per channel, median of each phase:")
for c in range(ch):
    sel = ok & (np.arange(len(t)) % ch == c)
    print("   ch %d " % c + " ".join("%6.0f" % np.median(d[sel, i]) for i in range(12)) + "   place wait %6.0f" % np.median(pl[sel]))
fr = np.arange(st.shape[0])
for c in range(ch):
    print("   ch %d start minus ch 0 start, frames 100/500/1000/2000: " % c, [int(st[k, c] - st[k, 0]) for k in (100, 500, 1000, 2000)])
print("   frame start (ch 0) minus first frame's, frames 100/500/1000/2000/2800:", [int(st[k, 0] - st[0, 0]) for k in (100, 500, 1000, 2000, 2800)])
for c in range(ch):
    sel = ok & (np.arange(len(t)) % ch == c)
    print("  channel %d: gather phase %6.0f  exchange phase %6.0f  total %6.0f" % (c, d[sel, 4].mean(), d[sel, 9].mean(), (t[sel, 12] - t[sel, 0]).mean()))
enc.close()
