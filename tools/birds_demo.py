import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import lbaudiodetective_amd as lb
B = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests/golden/birds")
for mode in (0, 1):
    d = lb.Detective(); d.set_file_hop_mode(mode)
    print("hop mode", mode, "same %.4f other %.4f self %.4f" % (
        d.compare_audio_urls(B + "/BlackBird.caf", B + "/BlackBird_eql.caf"),
        d.compare_audio_urls(B + "/BlackBird.caf", B + "/Sparrow_eql.caf"),
        d.compare_audio_urls(B + "/BlackBird.caf", B + "/BlackBird.caf")))
