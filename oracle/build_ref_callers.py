#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (checker side only; nothing under talkshow_amd/, nets/ or evaluation/ may import this).

Compiles the reference's own CALLER code — the functions of `scripts/demo.py` and `scripts/test_body.py` that drive the
`nets` package, plus the two helper modules they import from the reference tree — into `oracle/_ref/<unit>.code` + `oracle/_ref/reference_callers.json`
(code objects, the Python analogue of a compiled `oracle/_ref/*.so`, and a manifest of sha256 hashes; the directory is
git-ignored and travels to the GPU box with the snapshot).  `tests/test_reference_callers.py` executes those code objects against THIS repository's `nets`,
`evaluation` and SMPL-X layer on the GPU: the reference's callers driven against the drop-in, not an imitation of their call
shapes.  No reference source text is written anywhere (which is why the units are code objects and not `.py` files: the manifest
names file, lifted names and hashes so that what runs can be checked against the reference tree it came from).  The files are
raw `marshal` of ONE code object each — no pickle — and `load()` verifies every hash before unmarshalling anything.

What is lifted (by AST, so that the scripts' argument parsing, dataset and renderer imports stay out; the four `nets/spg` files of the
body path whole — `load_reference_modules()` — for bench.py's `cpu_baseline` kind "reference" and tests/test_reference_modules.py):
    scripts/demo.py        init_model (:30-64), infer (:158-247), the module-level `device` / `global_orient` assignments
    scripts/test_body.py   init_model (:30-56), body_loss (:98-110), test (:113-194)
    scripts/test_face.py   init_model (:25-53), face_loss (:78-95), test (:98-150)
    scripts/test_vq.py     test (:27-66)
    scripts/continuity.py  infer (:31-140), the module-level `global_orient`
    scripts/diversity.py   init_model (:30-64), get_vertices (:122-152), infer (:158-294), `global_orient`
    data_utils/lower_body.py, data_utils/get_j.py   whole modules (they import numpy / torch only)

    python oracle/build_ref_callers.py            # needs /root/reference; __graft_entry__.build() runs it where that exists
                                                  # (TS_SKIP_REF_CALLERS=1 opts out; a failure there is a warning + skipped tests)
"""
import ast
import hashlib
import json
import marshal
import os
import sys

REF = os.environ.get("TALKSHOW_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
MANIFEST = os.path.join(OUT_DIR, "reference_callers.json")

UNITS = {   # unit -> (file, names to keep: None = the whole module; functions and top-level assignments by name)
    "demo": ("scripts/demo.py", ["init_model", "infer", "device", "global_orient"]),
    "test_body": ("scripts/test_body.py", ["init_model", "body_loss", "test"]),
    "test_face": ("scripts/test_face.py", ["init_model", "face_loss", "test"]),
    "test_vq": ("scripts/test_vq.py", ["test"]),
    "continuity": ("scripts/continuity.py", ["infer", "global_orient"]),
    "diversity": ("scripts/diversity.py", ["init_model", "get_vertices", "infer", "global_orient"]),
    "lower_body": ("data_utils/lower_body.py", None),
    "get_j": ("data_utils/get_j.py", None),
    # the reference's own nn.Modules of the body path, whole files: bench.py's cpu_baseline (kind "reference") times THESE on the GPU
    # box's host cores, and tests/test_reference_modules.py checks them against the committed goldens
    "spg_gated_pixelcnn_v2": ("nets/spg/gated_pixelcnn_v2.py", None),
    "spg_vqvae_modules": ("nets/spg/vqvae_modules.py", None),
    "spg_wav2vec": ("nets/spg/wav2vec.py", None),
    "spg_vqvae_1d": ("nets/spg/vqvae_1d.py", None),
}
SPG_PACKAGE = "_talkshow_reference_spg"      # synthetic package the spg_* units are executed in (their relative imports resolve inside it)


class RefCallersError(RuntimeError):
    """The lift could not be made (reference tree absent or laid out differently) or a built file fails its integrity check."""


def _sha(b):
    return hashlib.sha256(b).hexdigest()


def lift(path, names):
    try:
        src = open(path, encoding="utf-8").read()
    except OSError as e:
        raise RefCallersError(f"cannot read {path}: {e}")
    try:
        tree = ast.parse(src, filename=path)
    except (SyntaxError, ValueError) as e:      # a reference file this interpreter cannot parse costs the caller tests, not the build
        raise RefCallersError(f"cannot parse {path}: {e}")
    if names is not None:
        keep = []
        for node in tree.body:
            if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
                keep.append(node)
            elif isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in node.targets):
                keep.append(node)
        found = {n.name for n in keep if hasattr(n, "name")} | {t.id for n in keep if isinstance(n, ast.Assign) for t in n.targets
                                                                 if isinstance(t, ast.Name)}
        missing = [n for n in names if n not in found]
        if missing:
            raise RefCallersError(f"{path}: not found: {missing} (the reference's layout changed?)")
        tree = ast.Module(body=keep, type_ignores=[])
    rel = os.path.relpath(path, REF)
    try:
        return compile(tree, f"<reference {rel}>", "exec"), _sha(src.encode())
    except (SyntaxError, ValueError, TypeError) as e:
        raise RefCallersError(f"cannot compile the lifted part of {path}: {e}")


def build(out_dir=OUT_DIR):
    """Writes one `<unit>.code` file (a marshalled code object — nothing else, no pickle) per unit and a JSON manifest that names,
    for every unit, the reference file it came from, that file's sha256, the names lifted and the sha256 of the `.code` bytes.
    Raises RefCallersError where the reference tree is absent or laid out differently."""
    if not os.path.isdir(REF):
        raise RefCallersError(f"{REF} does not exist: the reference callers can only be built where the reference tree is")
    os.makedirs(out_dir, exist_ok=True)
    manifest = {"python": list(sys.version_info[:2]), "units": {}}
    for unit, (rel, names) in UNITS.items():
        code, src_sha = lift(os.path.join(REF, rel), names)
        blob = marshal.dumps(code)
        with open(os.path.join(out_dir, unit + ".code"), "wb") as f:
            f.write(blob)
        manifest["units"][unit] = {"file": rel, "names": names, "source_sha256": src_sha, "code_sha256": _sha(blob),
                                   "code_bytes": len(blob)}
    with open(os.path.join(out_dir, "reference_callers.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    stale = os.path.join(out_dir, "reference_callers.bin")          # the pickle of earlier rounds
    if os.path.exists(stale):
        os.remove(stale)
    return os.path.join(out_dir, "reference_callers.json")


def load(out_dir=OUT_DIR):
    """-> ({unit: code object}, manifest), or None if nothing was built here / it was built by another Python version.

    Every `.code` file must hash to what the manifest says (and, where the reference tree is present, every reference file
    to the hash it had when lifted) before a single byte of it is unmarshalled: a file that fails raises RefCallersError, it is
    never executed.  This is a CORRUPTION / STALENESS check, not a defence against tampering — the manifest sits beside the files
    it describes, whoever can rewrite one can rewrite the other; the code objects are test infrastructure built from the local
    reference tree and executed by tests only.  The unit set and file names are fixed by UNITS above, not by the manifest."""
    path = os.path.join(out_dir, "reference_callers.json")
    if not os.path.exists(path):
        return None
    try:
        m = json.load(open(path))
    except (OSError, ValueError) as e:
        raise RefCallersError(f"{path} is unreadable: {e}")
    if list(sys.version_info[:2]) != m.get("python"):
        return None
    units = {}
    for unit, (rel, names) in UNITS.items():
        ent = m.get("units", {}).get(unit)
        if not ent or ent.get("file") != rel or ent.get("names") != names:
            raise RefCallersError(f"{path}: entry for unit {unit!r} does not match build_ref_callers.UNITS")
        try:
            blob = open(os.path.join(out_dir, unit + ".code"), "rb").read()
        except OSError as e:
            raise RefCallersError(f"{unit}.code named by the manifest is unreadable: {e}")
        if len(blob) != ent.get("code_bytes") or _sha(blob) != ent.get("code_sha256"):
            raise RefCallersError(f"{unit}.code does not hash to the manifest's code_sha256: refusing to load it")
        ref_file = os.path.join(REF, rel)
        if os.path.exists(ref_file) and _sha(open(ref_file, "rb").read()) != ent.get("source_sha256"):
            raise RefCallersError(f"{rel} changed since the callers were lifted: rebuild (python oracle/build_ref_callers.py)")
        try:
            units[unit] = marshal.loads(blob)
        except (EOFError, ValueError, TypeError) as e:
            raise RefCallersError(f"{unit}.code does not unmarshal: {e}")
    return units, m


def load_reference_modules(out_dir=OUT_DIR):
    """The reference's `nets/spg/{gated_pixelcnn_v2, vqvae_modules, wav2vec, vqvae_1d}.py` as importable modules, executed from the
    verified code objects inside a synthetic package (`SPG_PACKAGE`), or None where nothing was built.  Third-party modules those
    files import at the top and never use on this path (`torchvision`, `matplotlib.pyplot`) are stubbed WHILE the units execute if
    they are not installed, and removed again.  -> types.SimpleNamespace(GatedPixelCNN, VQVAE, AudioEncoder, modules={...})."""
    import importlib.machinery
    import types
    got = load(out_dir)
    if got is None:
        return None
    units, _ = got
    if SPG_PACKAGE + ".vqvae_1d" in sys.modules:
        mods = {n: sys.modules[SPG_PACKAGE + "." + n] for n in ("gated_pixelcnn_v2", "vqvae_modules", "wav2vec", "vqvae_1d")}
    else:
        import transformers  # noqa: F401   (before any stub: SURVEY.md Appendix C item 1)
        stubs = []
        for name in ("torchvision", "torchvision.datasets", "torchvision.transforms", "matplotlib", "matplotlib.pyplot"):
            try:
                __import__(name)
            except Exception:                                            # noqa: BLE001 — absent or broken: unused on this path either way
                m = types.ModuleType(name)
                m.__spec__ = importlib.machinery.ModuleSpec(name, None)
                m.__path__ = []
                sys.modules[name] = m
                stubs.append(name)
        for name in stubs:                                               # `from torchvision import datasets, transforms` / `import matplotlib.pyplot as plt`
            parent, _, child = name.rpartition(".")
            if parent in sys.modules and child:
                setattr(sys.modules[parent], child, sys.modules[name])
        pkg = types.ModuleType(SPG_PACKAGE)
        pkg.__path__ = []
        pkg.__spec__ = importlib.machinery.ModuleSpec(SPG_PACKAGE, None, is_package=True)
        sys.modules[SPG_PACKAGE] = pkg
        mods = {}
        try:
            for n in ("gated_pixelcnn_v2", "vqvae_modules", "wav2vec", "vqvae_1d"):     # dependency order
                m = types.ModuleType(SPG_PACKAGE + "." + n)
                m.__package__ = SPG_PACKAGE
                m.__spec__ = importlib.machinery.ModuleSpec(m.__name__, None)
                sys.modules[m.__name__] = m
                setattr(pkg, n, m)
                exec(units["spg_" + n], m.__dict__)
                mods[n] = m
        except Exception as e:                                           # noqa: BLE001
            for n in list(sys.modules):
                if n == SPG_PACKAGE or n.startswith(SPG_PACKAGE + "."):
                    del sys.modules[n]
            raise RefCallersError(f"the reference's nets/spg modules do not execute here: {e!r}")
        finally:
            for name in stubs:
                sys.modules.pop(name, None)
    return types.SimpleNamespace(GatedPixelCNN=mods["gated_pixelcnn_v2"].GatedPixelCNN, VQVAE=mods["vqvae_1d"].VQVAE,
                                 AudioEncoder=mods["vqvae_1d"].AudioEncoder, AE=mods["vqvae_1d"].AE, modules=mods)


if __name__ == "__main__":
    print("wrote", build())
