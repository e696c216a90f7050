"""The wave-level 512-point FFT of csrc/mel_fft.h (two real frames per complex transform, three radix-8 Stockham
stages, the lane/register pairing for Z[N-k]) emulated lane by lane on the host and checked against a direct DFT:
pins the index algebra of the mel kernel without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wave_fft_index_algebra_on_host(tmp_path):
  exe = str(tmp_path / 'mel_fft_check')
  subprocess.check_call(['g++', '-O1', '-std=c++17', '-I' + os.path.join(ROOT, 'speecht_amd', 'csrc'),
                         os.path.join(ROOT, 'tests', 'host_cpp', 'mel_fft_check.cpp'), '-o', exe])
  r = subprocess.run([exe], capture_output=True, text=True)
  assert r.returncode == 0, r.stdout + r.stderr
  assert 'max_err_a' in r.stdout
