"""One-time helper: lifts ONE kernel (text, descriptor, metadata entry) out of a `hipcc -S --cuda-device-only` dump into a
self-contained .s under a plain C name -- the starting point of a hand-maintained assembly kernel
(cusmc_amd/csrc/kernels/logpdf_nb4_gfx950.s was started this way and edited by hand from there).
    python scripts/extract_kernel_asm.py dump.s <mangled-name> <new-name> > out.s"""
import re
import sys

path, name, new = sys.argv[1:4]
s = open(path).read()
# ---- text: from the symbol's label to its .end_amdhsa_kernel + .size line
start = s.index("\n" + name + ":")
end_kernel = s.index(".end_amdhsa_kernel", start)
func_end = re.search(r"\.Lfunc_end\d+:\n\t\.size[^\n]*\n", s[end_kernel:])
body = s[start + 1:end_kernel + func_end.end()]
# drop the section switches inside (we emit our own), keep labels / instructions / descriptor
body = re.sub(r"\t\.section[^\n]*\n", "", body)
body = body.replace(name, new)
# local labels: make them unique to this file
body = re.sub(r"\.LBB\d+_(\d+)", r".L\1", body)
body = re.sub(r"\.Lfunc_end\d+", ".Lfunc_end_" + new, body)
# ---- metadata entry
md = s[s.index(".amdgpu_metadata"):]
k = md.index(".name:           " + name + "\n")
st = md.rfind("  - .agpr_count", 0, k)
nxt = md.find("  - .agpr_count", k)
entry = md[st:nxt if nxt >= 0 else md.index("amdhsa.target")]
entry = entry.replace(name, new)
print('\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"')
print("\t.text")
print("\t.protected\t%s" % new)
print("\t.globl\t%s" % new)
print("\t.p2align\t8")
print("\t.type\t%s,@function" % new)
text, rest = body.split("\t.amdhsa_kernel", 1)
print(text, end="")
print("\t.section\t.rodata,\"a\",@progbits")
print("\t.p2align\t6, 0x0")
desc, tail = rest.split(".end_amdhsa_kernel", 1)
print("\t.amdhsa_kernel" + desc + ".end_amdhsa_kernel")
print("\t.text")
print(tail, end="")
print("\t.amdgpu_metadata\n---\namdhsa.kernels:")
print(entry, end="")
print("amdhsa.target:   amdgcn-amd-amdhsa--gfx950\namdhsa.version:\n  - 1\n  - 2\n...\n\t.end_amdgpu_metadata")
