"""Put the @@...@@ placeholders of tools/fill_numbers.py back into DESIGN.md / README.md (the fill replaces them in place):
   python tools/retemplate_numbers.py <commit that holds the placeholders> <commit right after the fill>
Every line that differs between the two commits only by its filled figures is restored in the working copy if it is still there."""
import difflib, subprocess, sys
tmpl_commit, filled_commit = sys.argv[1], sys.argv[2]
for f in ("DESIGN.md", "README.md"):
    t = subprocess.check_output(["git", "show", tmpl_commit + ":" + f]).decode().split("\n")
    g = subprocess.check_output(["git", "show", filled_commit + ":" + f]).decode().split("\n")
    cur = open(f).read()
    n, miss = 0, []
    for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, t, g, autojunk=False).get_opcodes():
        if tag == "replace" and i2 - i1 == j2 - j1:
            for a, b in zip(t[i1:i2], g[j1:j2]):
                if "@@" in a:
                    if b in cur:
                        cur = cur.replace(b, a, 1); n += 1
                    else:
                        miss.append(a[:120])
        elif tag != "equal":
            miss += ["UNALIGNED " + a[:120] for a in t[i1:i2] if "@@" in a]
    open(f, "w").write(cur)
    print(f, "restored", n, "lines; not found:", len(miss))
    for m in miss:
        print("   ", m)
