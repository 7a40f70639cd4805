"""Per-kernel register / scratch / LDS report from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/kernel_resources.py <remarks file> [name filter]"""
import re
import subprocess
import sys


def parse(text):
    rows = []
    for blk in re.split(r"remark: Function Name: ", text)[1:]:
        name = blk.split()[0]

        def g(k):
            m = re.search(re.escape(k) + r": (\d+)", blk)
            return int(m.group(1)) if m else 0
        rows.append(dict(name=name, vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("TotalSGPRs"), scratch=g("ScratchSize [bytes/lane]"),
                         occ=g("Occupancy [waves/SIMD]"), lds=g("LDS Size [bytes/block]"), sgpr_spill=g("SGPRs Spill"),
                         vgpr_spill=g("VGPRs Spill")))
    return rows


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return [re.sub(r"\(.*", "", n).replace("void cnsn::", "") for n in out.splitlines()]


if __name__ == "__main__":
    rows = parse(open(sys.argv[1]).read())
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for r, dn in zip(rows, demangle([r["name"] for r in rows])):
        if flt in dn:
            print(f"{dn:78s} VGPR {r['vgpr']:4d} AGPR {r['agpr']:3d} SGPR {r['sgpr']:4d} scratch {r['scratch']:5d} "
                  f"sgpr-spill {r['sgpr_spill']:4d} vgpr-spill {r['vgpr_spill']:4d} occ {r['occ']} LDS {r['lds']}")
