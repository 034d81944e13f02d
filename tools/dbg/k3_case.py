import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import fastf_amd as F
n, n_groups, seed = 200_000, 7, 3
cells = np.arange(1, 1001, dtype=np.uint64) | (np.uint64(1) << np.uint64(62))
feats = np.arange(1, 501, dtype=np.uint64) | (np.uint64(2) << np.uint64(62))
eng = F.Engine(cells, feats, umi_max_bases=12)
rng = np.random.default_rng(seed)
fs, cs = 27, 36
cell = rng.integers(1, 1001, size=n_groups, dtype=np.uint64)
feat = rng.integers(1, 501, size=n_groups, dtype=np.uint64)
g = rng.integers(0, n_groups, size=n)
nonnull = (rng.random(n) > 0.1).astype(np.uint64)
umi = rng.integers(0, 64, size=n, dtype=np.uint64) * nonnull
ln = np.uint64(3) * nonnull
keys = (cell[g] << np.uint64(cs)) | (feat[g] << np.uint64(fs)) | (nonnull << np.uint64(26)) | (umi << np.uint64(2)) | ln
keys = np.sort(keys)
t = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
d_keys = t(keys); d_n = torch.tensor([n], dtype=torch.int64, device="cuda")
d_f = torch.zeros(n, dtype=torch.int32, device="cuda"); d_c = torch.zeros_like(d_f); d_k = torch.zeros_like(d_f)
d_nnz = torch.zeros(1, dtype=torch.int64, device="cuda")
for rep in range(3):
    eng.dev_reduce(d_keys.data_ptr(), d_n.data_ptr(), n, d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), d_nnz.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    nnz = int(d_nnz.item())
    grp = keys >> np.uint64(fs)
    ug, start = np.unique(grp, return_index=True)
    uk = np.unique(keys[(keys >> np.uint64(26)) & np.uint64(1) == 1])
    want = np.zeros(len(ug), dtype=np.int64); np.add.at(want, np.searchsorted(ug, uk >> np.uint64(fs)), 1)
    print("nnz", nnz, "want", len(ug), "group starts", start.tolist(), "sizes", np.diff(np.append(start, n)).tolist())
    print(" got counts", d_k.cpu().numpy()[:nnz].tolist(), "want", want.tolist())
    print(" got cells", d_c.cpu().numpy()[:nnz].tolist(), "want", (ug >> np.uint64(cs - fs)).tolist())
    print(" err", eng.dev_error_bits())
eng.close()
