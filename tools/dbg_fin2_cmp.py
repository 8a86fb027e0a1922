import numpy as np
a = np.load("gpurun_out/fin_main.npz"); b = np.load("gpurun_out/fin_f2.npz")
for k in a.files:
    d = a[k] != b[k]
    print(k, a[k].shape, "diff", int(d.sum()))
    if d.any() and k.startswith("y"):
        idx = np.argwhere(d)
        print("  channels", sorted(set(idx[:, 1]))[:70]); print("  rows", sorted(set(idx[:, 2]))[:60]); print("  cols", sorted(set(idx[:, 3]))[:60])
        i = tuple(idx[0]); print("  first", i, a[k][i], b[k][i])
    elif d.any():
        idx = np.argwhere(d); print("  first", idx[:5].tolist(), [hex(int(a[k][tuple(j)])) for j in idx[:3]], [hex(int(b[k][tuple(j)])) for j in idx[:3]])
