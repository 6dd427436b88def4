"""Where does a phase of the persistent decoder-step kernel spend its time?  100 MHz time stamps of one workgroup
(vacnic_decoder_step_args.trace) at config 5's shape (BART-large decoder, 5 beams, S = 512), averaged per phase type."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vacnic_amd import generate as Gn
from vacnic_amd.config import bart_large_vit_l14
from vacnic_amd.training import build_models

cfg, vcfg = bart_large_vit_l14()
model, _, _ = build_models(cfg, vcfg, device="cuda", seed=42, init="device", with_guide=False)
model.eval()
R, nb, S, Tmax = 5, 5, 512, 50
g = torch.Generator().manual_seed(0)
enc_h = (torch.randn(1, S, cfg.d_model, generator=g) * 0.5).bfloat16().cuda()
mask = torch.ones(1, S, dtype=torch.uint8).cuda()
dec = Gn.CachedDecoder(model, R, S, Tmax, reorders=True)
L = dec.L
names = ["P1 LN+kvq", "P2 self-attn", "P3 out", "P4 LN+q", "P5 cross-attn", "P6 out", "P7 LN+fc1", "P8 fc2"]
with torch.no_grad():
    dec.begin(enc_h, mask, nb)
    ids = torch.randint(3, 50000, (R, 1), generator=g).cuda()
    for t in range(30):
        dec.step(ids, t)
    for wg in (0, 100, 255) if len(sys.argv) < 2 else (0,):
        dec.trace = torch.zeros((8 * L + 1) * 8, device="cuda", dtype=torch.int64)
        dec.trace_wg = wg
        dec.step(ids, 30)
        torch.cuda.synchronize()
        tr = dec.trace.cpu().view(-1, 8).double() / 100.0          # us
        if dec.slots is not None:
            raw = dec.trace.cpu().view(-1, 8)
            dw, dc = (raw[8 * L, 6] - raw[8 * L, 4]).item(), (raw[8 * L, 7] - raw[8 * L, 5]).item()
            print(f"shader clock during the kernel: {dc} cycles in {dw / 100.0:.1f} us = {dc / (dw / 100.0) / 1e3:.2f} GHz")
            print(f"workgroup {wg} (slot variant): kernel span {(tr[8 * L - 1, 3] - tr[0, 1]).item():.1f} us over {8 * L} phases")
            print("  phase            wait+gather   compute+pack    total   (attention phases: pair workgroups only)")
            for k in range(8):
                rows = [ph for ph in range(8, 8 * L) if ph % 8 == k and tr[ph, 3] > 0 and tr[ph, 1] > 0]
                if not rows:
                    continue
                prev = lambda ph: max(tr[q, 3].item() for q in range(max(0, ph - 3), ph))
                ga = sum(tr[ph, 1].item() - prev(ph) for ph in rows) / len(rows)
                co = sum((tr[ph, 3] - tr[ph, 1]).item() for ph in rows) / len(rows)
                ln = [ph for ph in rows if tr[ph, 2] > 0]
                lns = f"   (LayerNorm {sum((tr[ph, 2] - tr[ph, 1]).item() for ph in ln) / len(ln):.2f})" if ln else ""
                print(f"  {names[k]:14s} {ga:11.2f} {co:14.2f} {ga + co:8.2f}{lns}")
            continue
        print(f"workgroup {wg}: kernel span {(tr[8 * L - 1, 3] - tr[0, 1]).item():.1f} us over {8 * L} phases")
        print("  phase            stage   compute+store  store-ack  arrive+issue  barrier-wait   total")
        for k in range(8):
            rows = [ph for ph in range(8, 8 * L - 1) if ph % 8 == k]
            st = sum((tr[ph, 1] - tr[ph, 0]).item() for ph in rows) / len(rows) if k not in (1, 4) else float("nan")
            cs = sum((tr[ph, 3] - (tr[ph, 1] if k not in (1, 4) else tr[ph, 0])).item() for ph in rows) / len(rows)
            ack = sum((tr[ph, 4] - tr[ph, 3]).item() for ph in rows) / len(rows)
            arr = sum((tr[ph, 5] - tr[ph, 4]).item() for ph in rows) / len(rows)
            bw = sum((tr[ph + 1, 0] - tr[ph, 5]).item() for ph in rows) / len(rows)
            tot = sum((tr[ph + 1, 0] - tr[ph, 0]).item() for ph in rows) / len(rows)
            print(f"  {names[k]:14s} {st:7.2f} {cs:12.2f} {ack:11.2f} {arr:12.2f} {bw:12.2f} {tot:9.2f}")
    dec.trace = None
    dec.check_step_kernel()
