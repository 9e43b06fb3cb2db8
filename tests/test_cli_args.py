"""The `metheor` executable's command-line contract, restating the reference's
tests/cli_error_handling.rs and tests/*-cli.rs cases that do not need a device (argument parsing,
usage errors, input-file errors happen before any GPU work).  Runs on CPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "metheor_amd", "metheor")
T1 = os.path.join(ROOT, "tests", "golden", "test1.bam")
MEASURES = ["pdr", "lpmd", "mhl", "pm", "me", "fdrp", "qfdrp"]


def run(*args, cwd=None):
    return subprocess.run([EXE, *args], capture_output=True, text=True, cwd=cwd or ROOT, timeout=120)


@pytest.fixture(scope="module", autouse=True)
def built():
    from metheor_amd import build
    build.build()
    assert os.path.exists(EXE)


def test_help_output():                      # cli_error_handling.rs:11-25
    r = run("--help")
    assert r.returncode == 0
    for s in ("Usage:", "Commands:", "pdr", "fdrp", "tag"):
        assert s in r.stdout


def test_version_output():                   # :27-37
    r = run("--version")
    assert r.returncode == 0 and "metheor" in r.stdout and "0.1.9" in r.stdout


def test_no_arguments_shows_help():          # :39-48
    r = run()
    assert r.returncode != 0 and "Usage:" in r.stderr


def test_invalid_subcommand():               # :50-61
    r = run("invalid_command")
    assert r.returncode != 0 and "error:" in r.stderr and "subcommand" in r.stderr


@pytest.mark.parametrize("sub", MEASURES)
def test_missing_required_args(sub, tmp_path):   # :64-134
    r = run(sub, "--output", str(tmp_path / "o.tsv"))
    assert r.returncode != 0 and "required" in r.stderr and "--input" in r.stderr
    r = run(sub, "--input", T1)
    assert r.returncode != 0 and "required" in r.stderr and "--output" in r.stderr


def test_invalid_min_depth_negative(tmp_path):   # :137-152
    r = run("pdr", "--input", T1, "--output", str(tmp_path / "o.tsv"), "--min-depth", "-5")
    assert r.returncode != 0 and ("invalid" in r.stderr or "error" in r.stderr)


def test_invalid_quality_threshold_out_of_range(tmp_path):   # :155-170 (u8 overflow)
    r = run("pdr", "--input", T1, "--output", str(tmp_path / "o.tsv"), "--min-qual", "300")
    assert r.returncode != 0 and "300" in r.stderr and "--min-qual" in r.stderr


def test_invalid_bam_file_format(tmp_path):      # :214-227
    q = tmp_path / "Cargo.toml"
    q.write_text("[package]\nname = \"metheor\"\n")
    r = run("pdr", "--input", str(q), "--output", str(tmp_path / "o.tsv"))
    assert r.returncode != 0 and "Error opening BAM file" in r.stderr


@pytest.mark.parametrize("sub", MEASURES)
def test_input_bam_doesnt_exist(sub, tmp_path):  # tests/{pdr,lpmd,mhl,me,pm,fdrp,qfdrp}-cli.rs:19-33
    r = run(sub, "-i", "tests/no_such.bam", "-o", str(tmp_path / "o.tsv"))
    assert r.returncode != 0 and "file not found" in r.stderr and "no_such.bam" in r.stderr


def test_nonexistent_cpg_set_file(tmp_path):     # :230-243
    r = run("pdr", "--input", T1, "--output", str(tmp_path / "o.tsv"), "--cpg-set", "nonexistent.bed")
    assert r.returncode != 0 and "Could not read target CpG file" in r.stderr


def test_tag_missing_reference(tmp_path):        # :246-259
    r = run("tag", "--input", T1, "--output", str(tmp_path / "o.sam"))
    assert r.returncode != 0 and "required" in r.stderr


# ---- tests/tag-cli.rs (the three failure cases need no device: they end before the first kernel) ----
def test_tag_input_file_doesnt_exist(tmp_path):  # tag-cli.rs:7-23
    r = run("tag", "-i", "tests/no_such_input.bam", "-o", str(tmp_path / "out.bam"), "-g", "tests/hg38.chr19.fa")
    assert r.returncode != 0 and "file not found" in r.stderr and "no_such_input.bam" in r.stderr


def test_tag_output_directory_doesnt_exist(tmp_path):   # tag-cli.rs:24-40, tag.rs:396-403
    sam = os.path.join(ROOT, "tests", "golden", "test.chr19.XM.sam")
    r = run("tag", "-i", sam, "-o", "no_such_dir/out.bam", "-g", "tests/hg38.chr19.fa")
    assert r.returncode == 101 and "No such directory" in r.stderr and "no_such_dir" in r.stderr
    # PathBuf::from("out.sam").parent() is "" -- not a directory: the reference refuses a bare file name too
    r = run("tag", "-i", sam, "-o", "out_bare_name.sam", "-g", "tests/hg38.chr19.fa")
    assert r.returncode == 101 and "No such directory" in r.stderr
    assert not os.path.exists(os.path.join(ROOT, "out_bare_name.sam"))


def test_tag_reference_genome_doesnt_exist(tmp_path):   # tag-cli.rs:41-58; tag.rs:386-390 error_when_reference_genome_is_not_found
    sam = os.path.join(ROOT, "tests", "golden", "test.chr19.XM.sam")
    out = tmp_path / "out.sam"
    r = run("tag", "-i", sam, "-o", str(out), "-g", "tests/no_such.fa")
    assert r.returncode == 101 and "file not found" in r.stderr and "no_such.fa" in r.stderr
    # the writer is created (and the header written) before the genome is opened (tag.rs:405-417)
    assert out.read_text().startswith("@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:chr19\tLN:58617616\n")


def test_subcommand_help_texts():                # :277-311
    r = run("pdr", "--help")
    assert r.returncode == 0 and "PDR" in r.stdout and "--input" in r.stdout and "--output" in r.stdout
    r = run("lpmd", "--help")
    assert r.returncode == 0 and "LPMD" in r.stdout and "--min-distance" in r.stdout and "--max-distance" in r.stdout
    r = run("tag", "--help")
    assert r.returncode == 0 and "Add bismark XM tag" in r.stdout and "--genome" in r.stdout


def test_zero_byte_input(tmp_path):              # :314-336
    q = tmp_path / "test_empty.bam"
    q.write_bytes(b"")
    r = run("pdr", "--input", str(q), "--output", str(tmp_path / "o.tsv"))
    assert r.returncode != 0


def test_defaults_match_lib_rs():
    """clap defaults of src/lib.rs:36-218 as printed by --help"""
    want = {"pdr": {"min-depth": "10", "min-cpgs": "4", "min-qual": "10"},
            "mhl": {"min-depth": "10", "min-cpgs": "4", "min-qual": "10"},
            "pm": {"min-depth": "10", "min-qual": "10"}, "me": {"min-depth": "10", "min-qual": "10"},
            "fdrp": {"min-qual": "10", "min-depth": "10", "max-depth": "40", "min-overlap": "35"},
            "qfdrp": {"min-qual": "10", "min-depth": "10", "max-depth": "40", "min-overlap": "35"},
            "lpmd": {"min-distance": "2", "max-distance": "16", "min-qual": "10"}}
    for sub, opts in want.items():
        out = run(sub, "--help").stdout
        for o, d in opts.items():
            line = [l for l in out.splitlines() if "--" + o + " " in l]
            assert line and "[default: %s]" % d in line[0], (sub, o, line)


def test_no_device_is_a_loud_failure(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = run("pdr", "-i", T1, "-o", str(tmp_path / "o.tsv"))
    assert r.returncode == 101 and "no CPU fallback" in r.stderr
    assert not (tmp_path / "o.tsv").exists()
