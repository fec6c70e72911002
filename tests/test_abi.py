"""The drop-in boundary: libmf.so loads, exports every symbol include/*.h declares and the five
C++ names an unchanged libphp_mf.so imports, and the reference's own unchanged
mfWarp.cpp links against it.  No compute calls: runs without a GPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def test_mfx_header_symbols_exported(pkg):
    src = open(os.path.join(INC, "mfx.h")).read()
    declared = set(re.findall(r"\b(mfx_[a-z0-9_]+)\s*\(", src))
    assert len(declared) >= 25
    missing = declared - exported(pkg.LIB_PATH)
    assert not missing, missing
    assert pkg.lib().mfx_abi_version() == 1


def test_mangled_facade_symbols_exported(pkg):
    syms = exported(pkg.LIB_PATH)
    for name, mangled in pkg.MANGLED.items():
        assert mangled in syms, name
    # the rest of include/mf.h
    for mangled in ["_ZN2mf8mf_trainEPKNS_10mf_problemENS_12mf_parameterE",
                    "_ZN2mf24mf_train_with_validationEPKNS_10mf_problemES2_NS_12mf_parameterE",
                    "_ZN2mf16mf_destroy_modelEPPNS_8mf_modelE", "_ZN2mf10mf_predictEPKNS_8mf_modelEii",
                    "_ZN2mf9calc_rmseEPNS_10mf_problemEPNS_8mf_modelE", "_ZN2mf20mf_get_default_paramEv",
                    "_ZN2mf12read_problemEPKc", "_ZN2mf13mf_save_modelEPKNS_8mf_modelEPKc"]:
        assert mangled in syms, mangled


def test_extern_c_shim_exports(pkg):
    src = open(os.path.join(INC, "mfwarp.h")).read()
    declared = set(re.findall(r"\b(php_[A-Za-z_]+)\s*\(", src))
    assert declared == {"php_mf_my_train", "php_utility_train", "php_utility_predict", "php_cos_similarity", "php_DINA"}
    assert declared <= exported(pkg.WARP_PATH)


def test_headers_compile_as_c_and_cpp(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "mfx.h"\n#include "mfwarp.h"\nint main(void){mfx_options o; mfx_default_options(&o); return sizeof(mfx_node)==12?0:1;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", INC, "-c", str(c), "-o", str(tmp_path / "t.o")])
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "mf.h"\nstatic_assert(sizeof(mf::mf_node)==12,"");\nint main(){mf::mf_parameter p=mf::mf_get_default_param(); return p.k;}\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", INC, "-c", str(cpp), "-o", str(tmp_path / "u.o")])


def test_reference_mfwarp_and_mftest_link_unchanged(pkg, tmp_path):
    """Compile the reference's own php_mf/mfWarp.cpp and mfTest/mfTest.cpp where they lie and link them
    against OUR libmf.so: every symbol they import must resolve (development container only)."""
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "php_mf", "mfWarp.cpp")):
        pytest.skip("reference sources not present")
    libdir = os.path.dirname(pkg.LIB_PATH)
    so = tmp_path / "libwarp_ref.so"
    subprocess.check_call(["g++", "-std=c++11", "-fPIC", "-shared", os.path.join(ref, "php_mf", "mfWarp.cpp"),
                           "-o", str(so), "-L", libdir, "-lmf", "-Wl,--no-undefined", "-Wl,-rpath," + libdir])
    assert {"php_utility_train", "php_utility_predict", "php_mf_my_train", "php_cos_similarity", "php_DINA"} <= exported(str(so))
    exe = tmp_path / "mfTest"
    subprocess.check_call(["g++", "-std=c++11", os.path.join(ref, "mfTest", "mfTest.cpp"), "-o", str(exe),
                           "-L", libdir, "-lmf", "-Wl,-rpath," + libdir])
    assert os.path.exists(exe)
