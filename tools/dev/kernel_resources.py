# dev: VGPR / AGPR / scratch / LDS of the library's kernels whose name contains one of the given substrings (the built .so is taken apart
# like tools/kernel_isa_pin.py does)     python tools/dev/kernel_resources.py ipa_scores node_tfmr
import sys, tempfile, subprocess, os, re
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import kernel_isa_pin as K
keys = sys.argv[1:] or [""]
for co in K.device_code_objects(K.LIB, tempfile.mkdtemp()):
    txt = subprocess.run([os.path.join(K.LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if not any(k in dn for k in keys):
            continue
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
        print("%-110s agpr %3s vgpr %3s sgpr %3s scratch %5s" % (dn.replace("(anonymous namespace)::", "")[:110], blk.split()[0], g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size")))
