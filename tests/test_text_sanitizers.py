"""The host-side glyph producer (csrc/text.cpp, csrc/path.cpp) under AddressSanitizer and UBSan, fed corrupted fonts."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FONT = os.path.join(ROOT, "contrast_renderer_amd", "data", "fonts", "OpenSans-Regular.ttf")


def test_corrupted_fonts_never_read_out_of_bounds():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "font_fuzz")
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"),
               os.path.join(ROOT, "tests", "cpp", "font_fuzz.cpp"), os.path.join(ROOT, "contrast_renderer_amd", "csrc", "text.cpp"),
               os.path.join(ROOT, "contrast_renderer_amd", "csrc", "path.cpp"), "-o", exe]
        build = subprocess.run(cmd, capture_output=True, text=True)
        assert build.returncode == 0, build.stderr[-2000:]
        run = subprocess.run([exe, FONT, "600"], capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
        parsed, rejected = (int(v) for v in run.stdout.split()[1::2])
        assert parsed > 100 and rejected > 100  # both outcomes occur; neither crashes
