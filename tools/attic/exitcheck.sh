# Which use of the library makes a python process crash at exit UNDER rocprofv3 (after the tool wrote its output)?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exitcheck
PRE="import numpy as np, sys, ctypes
sys.path.insert(0,'.')
from tostore_amd import HipVectorIndex, _ffi
L=_ffi.lib()
"
try() { # name, code
  timeout -k 5 120 rocprofv3 --kernel-trace -d gpurun_out/exitcheck/$1 -o x -- python -c "$PRE$2" > gpurun_out/exitcheck/$1.log 2>&1
  echo "$1 rc=$? segv=$(grep -c SIGSEGV gpurun_out/exitcheck/$1.log) abrt=$(grep -c "caught signal 6" gpurun_out/exitcheck/$1.log)"
}
try count "print(L.tsh_device_count())"
try create "idx = HipVectorIndex(64, 0); idx.close()"
try append "idx = HipVectorIndex(64, 0); idx.append(0, np.random.rand(5000,64).astype(np.float32)); idx.close()"
try search "idx = HipVectorIndex(64, 0); idx.append(0, np.random.rand(5000,64).astype(np.float32)); idx.search(np.random.rand(64).astype(np.float32), 5); idx.close()"
try batch "idx = HipVectorIndex(64, 0); idx.append(0, np.random.rand(5000,64).astype(np.float32)); idx.set_batch_min_nq(2); idx.search(np.random.rand(32,64).astype(np.float32), 5); idx.close()"
try noclose "idx = HipVectorIndex(64, 0); idx.append(0, np.random.rand(5000,64).astype(np.float32)); idx.search(np.random.rand(64).astype(np.float32), 5)"
export TSH_NO_CU_SPLIT=1
try create_nomask "idx = HipVectorIndex(64, 0); idx.close()"
unset TSH_NO_CU_SPLIT
try stream_only "import torch"
