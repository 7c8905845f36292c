"""The committed fixtures tests/golden/crosscheck_*.txt -- what tools/julia_crosscheck.jl replays inside ZigZagBoomerang.jl (the reference-side
pin of the oracle: tools/crosscheck/Project.toml, .github/workflows/crosscheck.yml) -- are byte for byte what the oracle produces TODAY: the
generator (tests/golden/export_crosscheck.py) is re-run into a scratch directory.  So a Julia run that passes pins the current oracle, not a
past one.  CPU only."""
import filecmp
import glob
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def test_every_crosscheck_fixture_is_regenerated_identically(tmp_path):
    spec = importlib.util.spec_from_file_location("export_crosscheck", os.path.join(HERE, "golden", "export_crosscheck.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.OUT = str(tmp_path)
    mod.main()
    mod.write_1d()
    mod.write_more()
    committed = sorted(glob.glob(os.path.join(HERE, "golden", "crosscheck_*.txt")))
    assert len(committed) >= 11
    for path in committed:
        fresh = os.path.join(str(tmp_path), os.path.basename(path))
        assert os.path.exists(fresh), os.path.basename(path)
        assert filecmp.cmp(path, fresh, shallow=False), os.path.basename(path)
