"""The C ABI: both shared libraries load on a machine without a GPU and export every entry point
that include/dib.h and include/dib_host.h declare (no compute calls here), the Python signature
tables cover exactly that set, and the product package never imports the oracle."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dib_[a-z0-9_]+)\s*\(", text)))


def test_device_library_exports_every_declared_symbol():
    from detectinblur_amd import _lib
    names = _declared("dib.h")
    assert len(names) >= 18
    l = _lib.lib()
    for n in names:
        assert getattr(l, n) is not None, n
    assert sorted(_lib.EXPORTS) == names          # every declared entry point has a ctypes signature, and nothing else
    assert l.dib_abi_version() == 7
    assert l.dib_tap_table_bytes(128) > 0 and l.dib_tap_table_bytes(100) == 0
    assert l.dib_tap_tables_bytes(128, 8) == 8 * l.dib_tap_table_bytes(128)
    assert l.dib_nms_workspace_bytes(128) == 128 * 2 * 8


def test_host_library_exports_every_declared_symbol():
    from detectinblur_amd import _hostlib
    names = _declared("dib_host.h")
    h = _hostlib.lib()
    for n in names:
        assert getattr(h, n) is not None, n
    assert set(_hostlib.EXPORTS) == set(names)


def test_argument_errors_are_reported_without_a_gpu():
    from detectinblur_amd import _lib
    l = _lib.lib()
    assert l.dib_psf_compact(None, 0, 1, 128, 1, None, None) == _lib.DIB_EINVAL
    assert b"null pointer" in l.dib_last_error()
    assert l.dib_nms(None, 70000, 0.5, None, None, None, None) == _lib.DIB_EINVAL
    assert b"16384" in l.dib_last_error()


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
    may touch it."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "detectinblur_amd")):
        for fn in files:
            if not fn.endswith(".py"):
                continue
            src = open(os.path.join(d, fn)).read()
            for node in ast.walk(ast.parse(src)):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom) and node.module:
                    mods = [node.module]
                if any(m.split(".")[0] in ("oracle", "dib_oracle", "ref_harness", "golden_inputs", "gen_goldens") for m in mods):
                    bad.append(os.path.join(d, fn))
            if "oracle" in src and ("sys.path" in src and "oracle" in src.split("sys.path")[1][:200]):
                bad.append(os.path.join(d, fn) + " (sys.path)")
    assert not bad, bad


def test_every_process_gets_a_private_copy_of_the_shipped_miopen_find_db():
    """`detectinblur_amd.use_shipped_kernel_choices()` -- NOT the import -- points MIOPEN_USER_DB_PATH at a temporary copy of
    detectinblur_amd/miopen_db (MIOpen appends to its user db; a shared one made kernel choices drift from process to process),
    respects a user's explicit setting, gives a child process a copy of its own, and works on the shipped directory itself only on
    request."""
    import subprocess
    import sys
    code = ("import os, detectinblur_amd; assert 'MIOPEN_USER_DB_PATH' not in os.environ or os.environ.get('DIB_TEST_PRESET'); "
            "detectinblur_amd.use_shipped_kernel_choices(); p = os.environ.get('MIOPEN_USER_DB_PATH'); print(p); "
            "print(sorted(os.listdir(p)) if p and os.path.isdir(p) else None)")
    shipped = os.path.join(ROOT, "detectinblur_amd", "miopen_db")
    files = sorted(f for f in os.listdir(shipped) if os.path.isfile(os.path.join(shipped, f)))
    assert files and all(f.endswith(".ufdb.txt") or f.endswith(".udb.txt") for f in files)       # find-db (+ the tuned perf-db)
    assert any(f.endswith(".ufdb.txt") for f in files)

    def run(extra):
        env = {k: v for k, v in os.environ.items() if k not in ("MIOPEN_USER_DB_PATH", "DIB_MIOPEN_DB_INPLACE", "DIB_NO_MIOPEN_DB",
                                                                "DIB_KERNEL_CHOICES_OWNER", "DIB_TEST_PRESET")}
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        path, listing = r.stdout.strip().splitlines()[-2:]
        return path, listing

    path, listing = run({})
    assert path != shipped and "dib_miopen_db_" in path and listing == str(files)
    assert not os.path.exists(path)                                   # removed when the process exits
    assert run({"MIOPEN_USER_DB_PATH": "/tmp", "DIB_TEST_PRESET": "1"})[0] == "/tmp"                       # the user's own setting
    # a path exported by THIS module in a parent process (marked by its pid) is not the user's: the child makes its own copy
    child = run({"MIOPEN_USER_DB_PATH": "/tmp/dib_miopen_db_of_a_parent", "DIB_KERNEL_CHOICES_OWNER": "1:miopen", "DIB_TEST_PRESET": "1"})
    assert "dib_miopen_db_" in child[0] and child[0] != "/tmp/dib_miopen_db_of_a_parent" and child[1] == str(files)
    assert run({"DIB_MIOPEN_DB_INPLACE": "1"})[0] == shipped
    assert run({"DIB_NO_MIOPEN_DB": "1"})[0] == "None"
    lines = sum(1 for _ in open(os.path.join(shipped, [f for f in files if f.endswith(".ufdb.txt")][0])))
    assert lines >= 500       # the bench's, the drivers' and the tests' shapes (`python -m detectinblur_amd.kernel_choices --fill --shapes bench`), not only the b = 8 training ones


def test_every_process_reads_a_private_copy_of_the_shipped_gemm_choices():
    """`detectinblur_amd.use_shipped_kernel_choices()` switches PyTorch's TunableOp to look-up-only mode on a temporary copy of
    detectinblur_amd/tunableop/tunableop_results.csv (one per device ordinal, as PyTorch names them), respects an explicit
    PYTORCH_TUNABLEOP_ENABLED and DIB_NO_TUNABLEOP, and the shipped file carries the validators PyTorch checks before using it."""
    import subprocess
    import sys
    code = ("import os, detectinblur_amd; detectinblur_amd.use_shipped_kernel_choices(); e = os.environ; f = e.get('PYTORCH_TUNABLEOP_FILENAME'); "
            "print(e.get('PYTORCH_TUNABLEOP_ENABLED'), e.get('PYTORCH_TUNABLEOP_TUNING'), f, "
            "len(os.listdir(os.path.dirname(f))) if f and os.path.isdir(os.path.dirname(f)) else None)")

    def run(extra):
        env = {k: v for k, v in os.environ.items() if not k.startswith("PYTORCH_TUNABLEOP") and k not in ("DIB_NO_TUNABLEOP", "DIB_KERNEL_CHOICES_OWNER")}
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout.strip().splitlines()[-1].split()

    on, tuning, path, n = run({})
    assert (on, tuning, n) == ("1", "0", "16") and "dib_tunableop_" in path and not os.path.exists(os.path.dirname(path))
    assert run({"DIB_NO_TUNABLEOP": "1"})[0] == "None"
    assert run({"PYTORCH_TUNABLEOP_ENABLED": "0"})[:3] == ["0", "None", "None"]
    shipped = open(os.path.join(ROOT, "detectinblur_amd", "tunableop", "tunableop_results.csv")).read().splitlines()
    validators = {l.split(",")[1] for l in shipped if l.startswith("Validator,")}
    assert {"PT_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION", "GCN_ARCH_NAME"} <= validators
    assert any("gfx950" in l for l in shipped) and sum(1 for l in shipped if l.startswith("Gemm")) >= 60


def test_kernel_choices_marker_names_only_the_variables_this_module_set(tmp_path):
    import subprocess
    import sys
    """ADVICE r5: a user's own MIOPEN_USER_DB_PATH must survive into grandchildren -- the owner marker lists the variable families
    this module exported, and a child overrides only those."""
    grandchild = tmp_path / "grandchild.py"
    grandchild.write_text("import os, sys\nsys.path.insert(0, %r)\nimport detectinblur_amd\ndetectinblur_amd.use_shipped_kernel_choices()\n"
                          "print(os.environ['MIOPEN_USER_DB_PATH'])\nprint(os.environ['PYTORCH_TUNABLEOP_FILENAME'])\n" % ROOT)
    child_script = tmp_path / "child.py"
    child_script.write_text("import os, json, subprocess, sys\nsys.path.insert(0, %r)\nimport detectinblur_amd\ndetectinblur_amd.use_shipped_kernel_choices()\n"
                            "own = os.environ['DIB_KERNEL_CHOICES_OWNER']\n"
                            "g = subprocess.run([sys.executable, %r], capture_output=True, text=True)\n"
                            "print(json.dumps([own, os.environ['MIOPEN_USER_DB_PATH'], os.environ.get('PYTORCH_TUNABLEOP_FILENAME'), g.stdout.split()]))\n"
                            % (ROOT, str(grandchild)))
    mine = str(tmp_path / "my_miopen_db")
    os.makedirs(mine)
    env = {k: v for k, v in os.environ.items() if k not in ("DIB_KERNEL_CHOICES_OWNER", "MIOPEN_USER_DB_PATH") and not k.startswith("PYTORCH_TUNABLEOP")}
    env["MIOPEN_USER_DB_PATH"] = mine
    import json
    p = subprocess.run([sys.executable, str(child_script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    own, parent_db, parent_csv, child = json.loads(p.stdout.strip().splitlines()[-1])
    assert own.split(":")[1] == "tunableop"                     # the find-db variable was the user's: not claimed
    assert parent_db == mine and child[0] == mine               # ... and it reaches the grandchild untouched
    assert child[1] != parent_csv and "dib_tunableop_" in child[1]    # the TunableOp copy WAS this module's: the child made its own


def test_kernel_choices_fill_tool_plumbing(tmp_path):
    import sys
    """`python -m detectinblur_amd.kernel_choices --fill`: a fresh child per shape set whose MIOpen user db and TunableOp file live
    in the output directory, seeded from the package, installed on request.  The worker is a stub here (no GPU)."""
    from detectinblur_amd import kernel_choices as kc
    pkg = tmp_path / "pkg"
    (pkg / "miopen_db").mkdir(parents=True)
    (pkg / "tunableop").mkdir()
    (pkg / "miopen_db" / "old.ufdb.txt").write_text("old\n")
    (pkg / "tunableop" / "tunableop_results.csv").write_text("Validator,PT_VERSION,x\n")
    stub = tmp_path / "stub.py"
    stub.write_text("import os, sys\n"
                    "assert os.environ['DIB_NO_MIOPEN_DB'] == '1' and os.environ['PYTORCH_TUNABLEOP_TUNING'] == '1' and 'DIB_KERNEL_CHOICES_OWNER' not in os.environ\n"
                    "d = os.environ['MIOPEN_USER_DB_PATH']\n"
                    "open(os.path.join(d, 'new.ufdb.txt'), 'a').write(sys.argv[1] + os.environ.get('MIOPEN_FIND_ENFORCE', '-') + '\\n')\n"
                    "f = os.environ['PYTORCH_TUNABLEOP_FILENAME'].replace('.csv', '0.csv')\n"
                    "assert open(f).read().startswith('Validator')\n"
                    "open(f, 'a').write('GemmTunableOp,%s\\n' % sys.argv[1])\n")
    res = kc.fill(["bench", "coco-eval"], str(tmp_path / "out"), tune=True, install=True, package_dir=str(pkg),
                  worker_argv=lambda name, tune: [sys.executable, str(stub), name])
    assert res["returncodes"] == {"bench": 0, "coco-eval": 0}
    assert res["miopen"] == ["new.ufdb.txt", "old.ufdb.txt"] and res["installed_into"] == str(pkg)
    assert (pkg / "miopen_db" / "new.ufdb.txt").read_text() == "bench3\ncoco-eval3\n"
    assert (pkg / "tunableop" / "tunableop_results.csv").read_text() == "Validator,PT_VERSION,x\nGemmTunableOp,bench\nGemmTunableOp,coco-eval\n"
    assert kc.foreign_hint({"miopen_foreign_files": ["x.ufdb.txt"]}).endswith(kc.FILL_COMMAND)
    assert kc.foreign_hint({"miopen_foreign_files": [], "tunableop_validators_match": True}) is None
    assert set(kc.SHAPE_SETS) == {"bench", "coco-train", "coco-eval"}
