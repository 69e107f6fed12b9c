"""Round 6: where a workgroup of the stream forward spends its life on the headline's short documents.
s_memtime stamps per wave (attn_fwd_stream.hip TRACE) -> per-phase cycle statistics over all workgroups."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd import _C  # noqa: E402
from touchnet_amd import library as L  # noqa: E402

dev, bf = "cuda", torch.bfloat16


def docs(B, T, mean, seed=0):
    rng = np.random.RandomState(seed)
    out = np.zeros((B, T), dtype=np.int32)
    for b in range(B):
        t, d = 0, 1
        while t < T:
            n = max(1, int(rng.normal(mean, mean * 0.1)))
            out[b, t:t + n] = d
            t += n
            d += 1
    return torch.from_numpy(out).to(dev)


B, T, Nh, D = 1, 15872, 32, 128
hpw = int(os.environ.get("HPW", "1"))
q, k, v = [torch.randn(B, T, Nh, D, dtype=bf, device=dev) for _ in range(3)]
doc = docs(B, T, 790)
mask = F.build_packed_mask(doc)
nwg = ((Nh + hpw - 1) // hpw) * ((T + 127) // 128) * B
buf = torch.zeros(nwg * 4 * 64, dtype=torch.int64, device=dev)
_C.lib().tn_attn_set_fwd_schedule(2)
for _ in range(3):
    L.attn_fwd(q, k, v, mask.doc, mask.meta, D ** -0.5)
_C.lib().tn_attn_fwd_stream_trace(_C.ptr(buf))
L.attn_fwd(q, k, v, mask.doc, mask.meta, D ** -0.5)
torch.cuda.synchronize()
_C.lib().tn_attn_fwd_stream_trace(None)
_C.lib().tn_attn_set_fwd_schedule(-2)
t = buf.cpu().numpy().reshape(nwg, 4, 64).astype(np.int64)
t0 = t[:, :, 0].min()
life, pro, trips_wait, trips_bar, trips_cmp, ntrip, seam = [], [], [], [], [], [], []
starts, ends = [], []
for wg in range(nwg):
    for w in range(4):
        ev = t[wg, w]
        nz = np.nonzero(ev)[0]
        if len(nz) < 5:
            continue
        last = nz[-1]
        starts.append(ev[0] - t0)
        ends.append(ev[last] - t0)
        life.append(ev[last] - ev[0])
        # events: 0 entry, 1 ids, 2 range, 3 list, then per trip (landed, barrier, computed), then per head seam, exit
        nt = (last - 4 - hpw) // 3 if hpw == 1 else None
        if hpw == 1 and nt is not None and nt >= 1:
            pro.append([ev[1] - ev[0], ev[2] - ev[1], ev[3] - ev[2]])
            for i in range(nt):
                a, b_, c = ev[4 + 3 * i], ev[5 + 3 * i], ev[6 + 3 * i]
                prev = ev[3 + 3 * i] if i > 0 else ev[3]
                trips_wait.append(a - prev)
                trips_bar.append(b_ - a)
                trips_cmp.append(c - b_)
            ntrip.append(nt)
            seam.append([ev[4 + 3 * nt] - ev[3 + 3 * nt], ev[last] - ev[4 + 3 * nt]])
life = np.array(life)
print(f"hpw={hpw} workgroups={nwg} waves traced={len(life)}  kernel span {max(ends)} cycles (s_memtime ticks)")
print(f"wave life: mean {life.mean():.0f} p50 {np.median(life):.0f} p90 {np.percentile(life, 90):.0f}")
if pro:
    pro = np.array(pro)
    print("prologue (cycles, mean): entry->ids %.0f  ids->range %.0f  range->list %.0f (the first trip's wait-landed is the first K/V tile)" % tuple(pro.mean(0)))
    print(f"trips per wave: mean {np.mean(ntrip):.1f}")
    print(f"per trip (cycles, mean / p50 / p90): wait-landed {np.mean(trips_wait):.0f}/{np.median(trips_wait):.0f}/{np.percentile(trips_wait, 90):.0f}  "
          f"barrier {np.mean(trips_bar):.0f}/{np.median(trips_bar):.0f}/{np.percentile(trips_bar, 90):.0f}  "
          f"compute {np.mean(trips_cmp):.0f}/{np.median(trips_cmp):.0f}/{np.percentile(trips_cmp, 90):.0f}")
    seam = np.array(seam)
    print("epilogue (cycles, mean): last trip->seam done %.0f  seam->exit %.0f" % tuple(seam.mean(0)))
    tot = pro.sum(1).mean() + np.mean(ntrip) * (np.mean(trips_wait) + np.mean(trips_bar) + np.mean(trips_cmp)) + seam.sum(1).mean()
    print(f"sum of the means {tot:.0f} cycles per wave")
starts, ends = np.array(starts), np.array(ends)
span = max(ends)
for frac in (0.25, 0.5, 0.75, 0.9, 1.0):
    print(f"  by {frac:.2f} of the span: {np.mean(starts <= frac * span):.2f} of waves started, {np.mean(ends <= frac * span):.2f} finished")
