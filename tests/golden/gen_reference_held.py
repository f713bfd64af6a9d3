#!/usr/bin/env python3
"""Generates tests/golden/reference_held_digests.json: digests that the reference's OWN tests hold as literals for the JDK
primitive this path uses (java.security.MessageDigest, Checksum.java:46-60, 83-84).

Run in the authoring container (the reference tree is only available there):

    python tests/golden/gen_reference_held.py

The reference's tests of the checksum path itself hold no literal digest.  One other test of the reference does, for
the same primitive: hadoop-ozone/s3gateway/src/test/java/org/apache/hadoop/ozone/s3/signature/TestStringToSignProducer.java
asserts the last line of an AWS V4 string-to-sign, which StringToSignProducer.createSignatureBase computes as
hash(canonicalRequest) = lower-case hex of MessageDigest.getInstance("SHA-256") over the request's canonical form
(hadoop-ozone/s3gateway/src/main/java/org/apache/hadoop/ozone/s3/signature/StringToSignProducer.java:112-130, 149-152).
This script reads that test AS TEXT (headers put, signed headers, method, path, expected literal), rebuilds the canonical
request the way buildCanonicalRequest does (:160-207), and keeps the pair only if hashlib's SHA-256 of the rebuilt message
equals the literal -- which proves the message is the one the Java test hashed.  Stored: the message bytes (derived from
the test's literal inputs) and the literal digest.  Nothing of the reference's source text is stored.

MD5: no test of the reference holds an MD5 digest of known content (the 32-hex strings in the SCM container-report tests
are fixed labels, not digests of anything), so MD5 stays pinned by RFC 1321 and hashlib only.
"""
import hashlib
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REL = "hadoop-ozone/s3gateway/src/test/java/org/apache/hadoop/ozone/s3/signature/TestStringToSignProducer.java"
TEST = os.path.join("/root/reference", REL)


def java_string(expr):
    """Value of a Java string expression made of literals joined with '+'."""
    return "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', expr)).replace("\\n", "\n")


def main():
    src = open(TEST).read()
    headers = {}
    for key, value in re.findall(r'headers\.put\(\s*("(?:[^"\\]|\\.)*"|[A-Za-z_.]+)\s*,\s*("(?:[^"\\]|\\.)*")\s*\)', src):
        if key.startswith('"'):
            headers[java_string(key).lower()] = java_string(value)   # LowerCaseKeyStringMap
        elif key.endswith("ORIGINAL_CONTENT_TYPE"):
            original_content_type = java_string(value)
    # headers.fixContentType(): the content type the client sent takes the place of the one the preprocessor set
    headers["content-type"] = original_content_type
    auth = java_string(re.search(r'String authHeader =(.*?)headers\.put\("Authorization"', src, re.S).group(1))
    signed = re.search(r"SignedHeaders=([^,]+),", auth).group(1).strip()
    call = re.search(r"createSignatureBase\((.*?)\);", src, re.S).group(1)
    scheme, method, path = [java_string(a) for a in re.findall(r'"(?:[^"\\]|\\.)*"', call)][:3]
    line_no = next(i + 1 for i, l in enumerate(src.split("\n")) if re.search(r'"[0-9a-f]{64}"', l))
    expected = re.search(r'"([0-9a-f]{64})"', src.split("\n")[line_no - 1]).group(1)

    canonical_headers = "".join("%s:%s\n" % (h.lower(), headers[h.lower()]) for h in signed.split(";"))
    payload_hash = headers["x-amz-content-sha256"]
    canonical_request = "\n".join([method, path, "", canonical_headers, signed, payload_hash])   # empty query string
    got = hashlib.sha256(canonical_request.encode("utf-8")).hexdigest()
    if got != expected:
        raise SystemExit("rebuilt canonical request hashes to %s, the test holds %s" % (got, expected))
    out = [{
        "type": "SHA256",
        "held_by": "%s:%d" % (REL, line_no),
        "computed_in_the_reference_by": "StringToSignProducer.hash (StringToSignProducer.java:149-152): "
                                        "MessageDigest.getInstance(\"SHA-256\") over the canonical request",
        "message_utf8": canonical_request,
        "digest_hex": expected,
    }]
    with open(os.path.join(HERE, "reference_held_digests.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %d record(s); message of %d bytes" % (len(out), len(canonical_request)))


if __name__ == "__main__":
    main()
