"""Per-kernel resources (LDS bytes, SGPRs, VGPRs, spills) from the ISA listing hipcc --save-temps leaves behind:
    cd /tmp/rt && hipcc --offload-arch=gfx950 <Makefile flags> -c <file>.hip --save-temps && python kernel_resources.py *.s [filter]"""
import re
import subprocess
import sys


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    pat = (r'\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count: (\d+).*?'
           r'\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count: (\d+)')
    for m in re.finditer(pat, s, re.S):
        lds, name, sg, ss, vg, vs = m.groups()
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(.*", "", dn).replace("void cgs::", "")
        if flt in dn:
            print(f"{dn[:64]:64s} lds={lds:>6s} sgpr={sg:>3s} vgpr={vg:>3s} vspill={vs:>3s} sspill={ss}")


main()
