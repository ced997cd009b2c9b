import sys
for l in open(sys.argv[1]):
    if l.startswith("  wave") or "kernel" in l or "conv" in l: print(l.rstrip()[:210])
