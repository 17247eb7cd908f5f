"""The Rust shim crate (rust/contrast_renderer_hip) cannot be compiled in this image (no rustc / cargo), so what can be checked is
checked: its raw bindings are GENERATED from include/contrast_hip.h and must equal a fresh generation (no drift), every export of the
header appears in the extern block with the right arity, every struct has the header's fields in order, and the handwritten shim
only calls functions the header declares and keeps the reference's public signatures."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "contrast_renderer_hip")


def _header():
    text = open(os.path.join(ROOT, "include", "contrast_hip.h")).read()
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def test_ffi_rs_is_a_fresh_generation_of_the_header():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"]).returncode == 0, \
        "rust/contrast_renderer_hip/src/ffi.rs is stale: run python tools/gen_rust_ffi.py"


def test_every_export_and_struct_field_of_the_header_is_bound():
    header = _header()
    ffi = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    declared = {}
    for m in re.finditer(r"\b(crh_\w+)\s*\(([^;{]*?)\)\s*;", header[header.index('extern "C" {'):], flags=re.S):
        params = " ".join(m.group(2).split())
        declared[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    bound = {m.group(1): (0 if not m.group(2).strip() else m.group(2).count(":")) for m in re.finditer(r"pub fn (crh_\w+)\(([^)]*)\)", ffi)}
    assert set(declared) == set(bound), set(declared) ^ set(bound)
    assert all(declared[k] == bound[k] for k in declared), {k: (declared[k], bound[k]) for k in declared if declared[k] != bound[k]}
    for m in re.finditer(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", header, flags=re.S):
        names = []
        for decl in m.group(1).split(";"):
            for part in decl.split(","):
                part = part.strip()
                if part:
                    names.append(re.match(r".*?(\w+)\s*(\[\w+\])?$", part).group(1))
        block = re.search(r"pub struct %s \{(.*?)\n\}" % m.group(2), ffi, flags=re.S).group(1)
        assert re.findall(r"pub (\w+):", block) == names, m.group(2)
    config = re.search(r"pub struct crh_config \{(.*?)\n\}", ffi, flags=re.S).group(1)
    assert len(re.findall(r"pub \w+: u32", config)) == 7  # all seven Configuration fields of the C ABI


def test_the_shim_keeps_the_reference_signatures_and_calls_only_declared_functions():
    lib = open(os.path.join(CRATE, "src", "lib.rs")).read()
    ffi = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    bound = set(re.findall(r"pub fn (crh_\w+)\(", ffi))
    called = set(re.findall(r"ffi::(crh_[a-z_0-9]+)\(", lib))
    assert called and called <= bound, called - bound
    # the four functions that cross the seam in the reference (renderer.rs:432, :177-183, :267-273, :360-376) and what they return
    assert re.search(r"pub fn new\(device: i32, config: Configuration\) -> Result<Self, Error>", lib)
    assert re.search(r"pub fn from_paths\(renderer: &Renderer, dynamic_stroke_options: &\[DynamicStrokeOptions\], paths: &\[Path\], existing_shape: Option<Shape>\) -> Result<Self, Error>", lib)
    assert re.search(r"pub fn render\(&self, _renderer: &Renderer, render_pass: &mut RenderPass, instance_indices: Range<u32>, render_operation: RenderOperation\)", lib)
    assert re.search(r"pub fn set_dynamic_stroke_options\(&self, dynamic_stroke_options_group_index: usize, dynamic_stroke_options_group: &DynamicStrokeOptions\) -> Result<\(\), Error>", lib)
    # Error variants in the reference's order (error.rs:5-16) = statuses 1..5
    body = re.sub(r"//[^\n]*", "", re.search(r"pub enum Error \{(.*?)\n\}", lib, flags=re.S).group(1))
    variants = body.replace(",", " ").split()
    # ... and nothing of the shim's own: clip / alpha state outlives a crh_scene_render_draws call (the frame keeps it), so a pass may span Shape objects
    assert variants == ["NumberOfStencilBitsIsUnsupported", "ClipStackOverflow", "TooManyNestedOpacityGroups", "TooManyDashIntervals", "DynamicStrokeOptionsIndexOutOfBounds"]
    assert "PassStateSpansShapes" not in lib
    # enum discriminants that cross the ABI as integers
    for name, items in (("SegmentType", ["Line = 0", "IntegralQuadraticCurve = 1", "IntegralCubicCurve = 2", "RationalQuadraticCurve = 3", "RationalCubicCurve = 4"]),
                        ("RenderOperation", ["Stencil = 0", "Clip = 1", "UnClip = 2", "Color = 3", "SaveAlphaContext = 4", "ScaleAlphaContext = 5", "RestoreAlphaContext = 6"]),
                        ("Cap", ["Square = 0", "Round = 1", "Out = 2", "In = 3", "Right = 4", "Left = 5", "Butt = 6"])):
        body = re.search(r"pub enum %s \{(.*?)\}" % name, lib, flags=re.S).group(1)
        assert [i.strip() for i in body.split(",") if i.strip()] == items, name
    assert os.path.exists(os.path.join(CRATE, "Cargo.toml")) and "cargo:rustc-link-lib=dylib=contrast_hip" in open(os.path.join(CRATE, "build.rs")).read()
    # braces balance (the nearest thing to a syntax check available here)
    for f in ("lib.rs", "ffi.rs"):
        text = re.sub(r"//.*", "", open(os.path.join(CRATE, "src", f)).read())
        text = re.sub(r'"(?:[^"\\]|\\.)*"', '""', text)
        assert text.count("{") == text.count("}") and text.count("(") == text.count(")") and text.count("[") == text.count("]"), f
