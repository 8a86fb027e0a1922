"""Static instruction mix of one kernel in a hipcc -save-temps .s file: python tools/isa_mix.py file.s substring [n]"""
import collections, sys
s = open(sys.argv[1]).read()
names = [l.split(":")[0] for l in s.split("\n") if ": " in l and "@" in l and sys.argv[2] in l.split(":")[0] and l[0] not in ".; \t"]
for name in names:
    i = s.index("\n" + name + ":")
    j = s.index(".Lfunc_end", i)
    c = collections.Counter()
    for l in s[i:j].split("\n")[2:]:
        l = l.strip()
        if not l or l.startswith((".", ";", "_")) or l.endswith(":"):
            continue
        c[l.split()[0]] += 1
    print(name, sum(c.values()))
    print("  " + "  ".join("%s %d" % kv for kv in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40)))
