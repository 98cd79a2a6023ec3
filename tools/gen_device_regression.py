# Writes the device-side regression fixtures of the beam searches (tests/golden/device_beam_*.npz) on the MI355X box:
#   /usr/local/graft/bin/gpurun -- 'python tools/gen_device_regression.py'   -> gpurun_out/regress/*.npz, then copy to tests/golden/
# It runs the two parity tests with GITCAP_WRITE_REGRESSION set, so the fixture is exactly what those tests compare.
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, GITCAP_WRITE_REGRESSION=os.path.join(root, "gpurun_out", "regress"))
sys.exit(subprocess.call([sys.executable, "-m", "pytest", "-q", "-s", "-m", "gpu", os.path.join(root, "tests", "test_parity_gpu.py"), "-k",
                          "test_config4_real_shape_fp8_beam or test_device_beam_search_base_size"], env=env, cwd=root))
