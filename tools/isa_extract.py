"""Cut one kernel out of a hipcc -save-temps .s file:  python tools/isa_extract.py file.s 'ILi3ELi3ELi2ELi8ELi1ELi4ELb0ELb0ELb1ELb0ELi1ELb0E' > out.s
Prints the body (labels + instructions, comments stripped) and, on stderr, per-basic-block instruction counts."""
import collections, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [l.split(":")[0] for l in s.split("\n") if l.startswith("_Z") and ":" in l and key in l.split(":")[0]]
for name in names[:1]:
    i = s.index("\n" + name + ":")
    j = s.index(".Lfunc_end", i)
    blk, counts, order = "entry", collections.OrderedDict(), []
    for l in s[i:j].split("\n")[2:]:
        t = l.split(";")[0].rstrip()
        if not t.strip():
            continue
        if t.strip().startswith("."):
            if t.strip().endswith(":"):
                blk = t.strip()[:-1]
                print(t)
            continue
        print(t)
        op = t.split()[0]
        c = counts.setdefault(blk, collections.Counter())
        cls = ("valu" if op.startswith("v_") and not op.startswith("v_mfma") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_nop")) else
               "smem" if op.startswith(("s_load", "s_buffer")) else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "lds" if op.startswith("ds_") else "other")
        c[cls] += 1
        if op in ("v_bcnt_u32_b32", "v_and_b32", "v_bitop3_b32"):
            c["pair"] += 1
    for b, c in counts.items():
        print(b, dict(c), file=sys.stderr)
