"""Does a 2-D torch gather t[idx] of more than 2^32 bytes come back right on this stack?  (DenseMatrix.__getitem__
with a row list on a 10 GB block, CsrDev.take_rows.)"""
import torch
n, k = 10_000_000, 128
t = torch.arange(n, device="cuda", dtype=torch.float64)[:, None] * 1000 + torch.arange(k, device="cuda", dtype=torch.float64)[None, :]
g = torch.Generator(device="cuda"); g.manual_seed(0)
idx = torch.randperm(n, device="cuda", generator=g)[:6_000_000]
sub = t[idx]
ok = bool((sub[:, 0] == idx.to(torch.float64) * 1000).all()) and bool((sub[:, 127] == idx.to(torch.float64) * 1000 + 127).all())
print("float64 (n, 128) row gather of 6.1 GB:", "OK" if ok else "WRONG")
sub2 = torch.index_select(t, 0, idx)
print("index_select:", "OK" if bool((sub2[:, 5] == idx.to(torch.float64) * 1000 + 5).all()) else "WRONG")
b = torch.arange(300_000_000 * 4, device="cuda", dtype=torch.int32).view(-1, 4)
perm = torch.randperm(300_000_000, device="cuda", generator=g)
a = b[perm]
print("int32 (3e8, 4) gather of 4.8 GB:", "OK" if bool((a[:, 0] == (perm * 4).to(torch.int32)).all() and (a[:, 3] == (perm * 4 + 3).to(torch.int32)).all()) else "WRONG")
a2 = torch.index_select(b, 0, perm)
print("index_select int32:", "OK" if bool((a2[:, 3] == (perm * 4 + 3).to(torch.int32)).all()) else "WRONG")
