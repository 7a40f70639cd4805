import sys
rows=[l.split("|") for l in open(sys.argv[1]) if l.startswith("| ") and l[2].isdigit()]
v=[(int(r[1]),r[2].strip(),int(r[3]),int(r[4]),int(r[5])) for r in rows]
def cls(c): return "G" if c>5700 else ("m" if c>5300 else ".")
print("write-target class by buffer #:")
print("".join(cls(x[2]) for x in v))
print("read-from-it class (>5700 G, >5300 m):")
print("".join(cls(x[4]) for x in v))
