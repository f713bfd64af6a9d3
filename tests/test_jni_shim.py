"""The JNI shim and its Java class cannot be built here (no JDK).  What can be checked without one: the shim is
valid C against the C ABI header (syntax and types, through a hand-written declaration stub of the JNI names it
uses), and every native method GpuChecksum.java declares has a definition in the shim with the same number of
parameters and matching JNI types."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "hadoop-ozone_amd", "jni", "ozcs_jni.c")
JAVA = os.path.join(ROOT, "hadoop-ozone_amd", "jni", "GpuChecksum.java")

JAVA_TO_JNI = {"int": "jint", "long": "jlong", "boolean": "jboolean", "byte[]": "jbyteArray", "int[]": "jintArray",
               "ByteBuffer": "jobject", "void": "void"}


def test_shim_is_valid_c_against_the_abi_header():
    r = subprocess.run(["gcc", "-fsyntax-only", "-std=gnu11", "-Wall", "-Wextra", "-Werror",
                        "-I", os.path.join(ROOT, "tests", "jni_stub"), "-I", os.path.join(ROOT, "include"), SHIM],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_shim_holds_no_critical_region_and_makes_no_upcalls():
    src = open(SHIM).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for banned in ("GetPrimitiveArrayCritical", "FindClass", "GetMethodID", "CallIntMethod"):
        assert banned not in code, banned


def test_every_java_native_method_has_a_matching_definition():
    java = open(JAVA).read()
    natives = re.findall(r"(?:public|private)\s+static\s+native\s+(\S+)\s+(\w+)\s*\(([^)]*)\)", java)
    assert len(natives) >= 10
    shim = re.sub(r"/\*.*?\*/", "", open(SHIM).read(), flags=re.S)
    defs = {m.group(2): (m.group(1), m.group(3)) for m in
            re.finditer(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+JFN\((\w+)\)\s*\(([^)]*)\)", shim)}
    assert set(defs) == {name for _, name, _ in natives}
    for ret, name, params in natives:
        jtypes = [JAVA_TO_JNI[" ".join(p.split()[:-1])] for p in params.split(",") if p.strip()]
        cret, cparams = defs[name]
        ctypes_ = [" ".join(p.split()[:-1]) for p in cparams.split(",")]
        assert ctypes_[:2] == ["JNIEnv*", "jclass"], name
        assert ctypes_[2:] == jtypes, (name, ctypes_[2:], jtypes)
        assert cret == JAVA_TO_JNI[ret], name
