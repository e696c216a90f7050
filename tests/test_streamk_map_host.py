"""The work partition of the persistent per-bin product kernels (csrc/streamk_map.h) walked exhaustively on the host:
every (tile, k-tile) unit exactly once, a tile's pieces on consecutive workgroups, every published piece the first piece of
its workgroup and as long as the head's owner predicts -- the index algebra and the hand-off pairing of the kernel, pinned without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_streamk_partition_on_host(tmp_path):
  exe = str(tmp_path / 'streamk_map_check')
  subprocess.check_call(['g++', '-O2', '-std=c++17', '-I' + os.path.join(ROOT, 'speecht_amd', 'csrc'),
                         os.path.join(ROOT, 'tests', 'host_cpp', 'streamk_map_check.cpp'), '-o', exe])
  r = subprocess.run([exe], capture_output=True, text=True)
  assert r.returncode == 0, r.stdout + r.stderr
  assert 'wgs_per_xcd=64 upw=18 / wgs_per_xcd=96 upw=12' in r.stdout, r.stdout
