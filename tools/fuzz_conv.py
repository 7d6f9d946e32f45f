"""Random-shape parity of the 3x3 conv forward / backward entry points (both sources, signed pad, masks) against fp64
torch on the CPU.  Run on the GPU box: python tools/fuzz_conv.py [n] [seed]."""
import sys, os, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dl-unet_amd"))
import torch, torch.nn.functional as F
import _hip
TOL = 2e-5
BF16 = os.environ.get("UNET_MATH") == "2"        # bf16 tensors: inputs rounded to bf16 first, activations judged at 4e-3, fp32 results at 2e-5
TOL_ACT = 4e-3 if BF16 else TOL

def nerr(a, ref):
    a = a.double().cpu(); ref = ref.double().cpu()
    return ((a - ref).abs().max() / ref.abs().max().clamp_min(1e-300)).item()
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16 if BF16 else torch.float32).cuda()
nchw = lambda t: t.permute(0, 3, 1, 2).double().cpu()


def run(n, seed, verbose=True):
  L = _hip.lib()
  rng = random.Random(seed)
  worst = 0.0
  for it in range(n):
      # (32 and 96: the 32-row filter blocks of the Winograd kernel and half-filled weight-gradient tiles; fp32 modes only -
      #  bf16 tensors need whole 64-channel tiles)
      B = rng.choice([1, 1, 2, 3]); C = rng.choice([64, 64, 128, 192, 256] + ([] if BF16 else [32, 96])); K = rng.choice([64, 128, 192, 256] + ([] if BF16 else [32, 96]))
      concat = rng.random() < 0.4
      if concat:
          Hs = rng.randint(8, 40); pad = rng.randint(-3, 7); H = Hs + 2 * pad
          if H < 6 or Hs < 4: continue
      else:
          H = rng.randint(6, 64); Hs, pad = H, 0
      g = torch.Generator().manual_seed(it)
      r = (lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).float().to(torch.bfloat16).double()) if BF16 else (lambda *s: torch.randn(*s, generator=g, dtype=torch.float64))
      a = r(B, C, Hs, Hs).requires_grad_(True)
      u = r(B, C, H, H).requires_grad_(True) if concat else None
      Ct = 2 * C if concat else C
      w = (r(K, Ct, 3, 3) * 0.05); w = (w.float().to(torch.bfloat16).double() if BF16 else w).requires_grad_(True); b = r(K)
      dz = r(B, K, H - 2, H - 2)
      xin = torch.cat((F.pad(a, (pad,) * 4), u), 1) if concat else a
      z = F.conv2d(xin, w, b); z.backward(dz)
      keep = []
      def k(t): keep.append(t); return t
      adt = torch.bfloat16 if BF16 else torch.float32
      y = torch.empty(B, H - 2, H - 2, K, device="cuda", dtype=adt)
      sc = torch.empty(L.unet_conv3x3_scratch_bytes(Ct, K), dtype=torch.uint8, device="cuda")
      _hip.check(L.unet_conv3x3_fwd(_hip.ptr(k(nhwc(a.detach()))), Hs, Hs, C, pad, _hip.ptr(k(nhwc(u.detach()))) if concat else None, C if concat else 0,
                                    B, H, H, _hip.ptr(k(w.detach().float().cuda())), _hip.ptr(k(b.float().cuda())), K, 1, _hip.ptr(y), _hip.ptr(sc), _hip.stream()))
      e_f = nerr(nchw(y), F.relu(z.detach()))
      dx1 = torch.empty(B, Hs, Hs, C, device="cuda", dtype=adt); dx2 = torch.empty(B, H, H, C, device="cuda", dtype=adt) if concat else None
      dw = torch.empty(K, Ct, 3, 3, device="cuda"); db = torch.empty(K, device="cuda")
      sc2 = torch.empty(L.unet_conv3x3_bwd_scratch_bytes(B, H, H, Ct, K), dtype=torch.uint8, device="cuda")
      _hip.check(L.unet_conv3x3_bwd(_hip.ptr(k(nhwc(a.detach()))), Hs, Hs, C, pad, _hip.ptr(k(nhwc(u.detach()))) if concat else None, C if concat else 0,
                                    B, H, H, _hip.ptr(k(w.detach().float().cuda())), K, _hip.ptr(k(nhwc(dz))), _hip.ptr(dx1), None, None,
                                    _hip.ptr(dx2) if concat else None, None, _hip.ptr(dw), _hip.ptr(db), _hip.ptr(sc2), _hip.stream()))
      errs = [e_f, nerr(nchw(dx1), a.grad), nerr(dw, w.grad), nerr(db, dz.sum((0, 2, 3)))]
      tols = [TOL_ACT, TOL_ACT, TOL, TOL]
      if concat: errs.append(nerr(nchw(dx2), u.grad)); tols.append(TOL_ACT)
      m = max(e / t for e, t in zip(errs, tols)) * TOL; worst = max(worst, m)
      if verbose: print("%2d B=%d C=%d K=%d Hs=%d pad=%d H=%d concat=%d: %s %s" % (it, B, C, K, Hs, pad, H, concat, " ".join("%.1e" % e for e in errs), "" if m < TOL else "  <-- FAIL"), flush=True)
  return worst


if __name__ == "__main__":
    w = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("worst", w, "OK" if w < TOL else "FAIL")
