"""Where a batch call on the sixty fixtures spends its HOST time: needs the experiment build of api_files.cpp

    cd lbaudiodetective_amd/csrc && hipcc <CXXFLAGS of the Makefile> -DLBAD_EXP_FILE_TIMES -x hip -c api_files.cpp -o ../lib/exp/api_files.ft.o
    hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib/exp/lib_ft.so <the other objects of ../lib/obj> ../lib/exp/api_files.ft.o -ldl

which prints, per run of files and per group, the microseconds of: sizes + plan, read + parse (+ the readers' uploads),
table building + launches, waiting for the device, unpacking (stderr).  Round 3, final build: 0.12 / 0.32 / 0.05 / 1.9 /
0.14 ms of a 2.5 ms call (before the converter's tap loop was rewritten)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from lbaudiodetective_amd import _native as N
N.LIB_PATH = os.path.join(root, "lbaudiodetective_amd", "lib", "exp", "lib_ft.so")
import lbaudiodetective_amd as lb
birds = os.path.join(root, "tests", "golden", "birds")
paths = sorted(os.path.join(birds, f) for f in os.listdir(birds) if f.endswith(".caf"))
det = lb.Detective()
for _ in range(4):
    det.process_audio_urls(paths)
