"""Which expression is torch.norm(d, dim=-1) of [N, 3] fp32 rows on this GPU? (tools/gpu: run on the box)"""
import torch
torch.manual_seed(0)
d = (torch.randn(1 << 20, 3, device="cuda") * torch.tensor([1.0, 0.7, 1.3], device="cuda")).float()
want = torch.norm(d, dim=-1)
x, y, z = (d[:, i].double() for i in range(3))
f32 = lambda t: t.float().double()
fma = lambda a, b, c: f32(a * b + c)
sq = lambda a: f32(a * a)
cands = {
    "sqrt((xx + yy) + zz)": f32(f32(sq(x) + sq(y)) + sq(z)),
    "sqrt(xx + (yy + zz))": f32(sq(x) + f32(sq(y) + sq(z))),
    "sqrt((xx + zz) + yy)": f32(f32(sq(x) + sq(z)) + sq(y)),
    "fma(z,z,fma(y,y,xx))": fma(z, z, fma(y, y, sq(x))),
    "fma(x,x,fma(y,y,zz))": fma(x, x, fma(y, y, sq(z))),
    "fma(z,z,xx+yy)": fma(z, z, f32(sq(x) + sq(y))),
    "fma(y,y,xx)+zz": f32(fma(y, y, sq(x)) + sq(z)),
    "fma(z,z,fma(x,x,yy))": fma(z, z, fma(x, x, sq(y))),
    "exact sum, one rounding": f32(x * x + y * y + z * z),
}
for name, s in cands.items():
    got = torch.sqrt(s.float())
    print(f"{name:28s} mismatches {(got != want).sum().item():8d}")
v = d / want[:, None]
print("division d / norm == torch.div:", torch.equal(v, torch.div(d, want[:, None])))
