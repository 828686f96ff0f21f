"""The host side of libanx (model, index build, confusables, context rules, index image, formatters, C ABI) under
AddressSanitizer + UndefinedBehaviorSanitizer: the host sources are compiled with g++ -fsanitize=address,undefined
against device stubs (tests/host_sanitize/stub_engine.cpp; GPU sanitizers are not available on the pool) and driven by
tests/host_sanitize/main.cpp on the golden lexicon."""
import os
import subprocess

from analiticcl_amd import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SOURCES = ["host_model.cpp", "capi.cpp", "search.cpp", "confusables.cpp", "contextrules.cpp", "index_cache.cpp", "adjacency.cpp"]


def test_host_code_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_sanitize"
    src = [os.path.join(REPO, "analiticcl_amd", "csrc", f) for f in HOST_SOURCES]
    src += [os.path.join(REPO, "tests", "host_sanitize", f) for f in ("stub_engine.cpp", "main.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-pthread", "-I", os.path.join(REPO, "include"), "-o", str(exe)] + src
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    data = synth.materialize_golden(str(tmp_path / "data"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([str(exe), data["alphabet"], data["eng"], str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert r.stdout.strip().startswith("OK ")
    # the multi-replica sharding of the batch calls (contiguous input ranges, one host thread per replica, rows concatenated in
    # input order) against the stub's fake devices: shard == whole for 1..4 replicas, every input form, confusables, search mode
    env["ANX_STUB_FAKE"] = "1"
    r = subprocess.run([str(exe), data["alphabet"], data["eng"], str(tmp_path), "shards"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert r.stdout.strip().startswith("OK ")


def test_host_threads_under_tsan(tmp_path):
    """The same harness under ThreadSanitizer, multi-replica mode (one host thread per replica, the host pool's loops, the threaded
    build of the signature adjacency lists, the length split's shared cost model): no data race is reported."""
    exe = tmp_path / "host_tsan"
    src = [os.path.join(REPO, "analiticcl_amd", "csrc", f) for f in HOST_SOURCES]
    src += [os.path.join(REPO, "tests", "host_sanitize", f) for f in ("stub_engine.cpp", "main.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=thread", "-pthread", "-I", os.path.join(REPO, "include"),
           "-o", str(exe)] + src
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    data = synth.materialize_golden(str(tmp_path / "data"))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1", ANX_STUB_FAKE="1", ANX_HARNESS_QUICK="1")
    r = subprocess.run([str(exe), data["alphabet"], data["eng"], str(tmp_path), "shards"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, (r.stdout[-2000:], r.stderr[-6000:])
    assert r.stdout.strip().startswith("OK ")
