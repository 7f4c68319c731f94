"""The `ffi.cdef` body of fluidnet_amd/lua/tfluids_hip.lua, generated from include/tfluids_hip.h: comments, preprocessor
lines and the extern "C" wrapper stripped, nothing else changed -- so the LuaJIT binding declares exactly what the header
declares. tests/test_lua_binding.py regenerates it and diffs it against the block embedded in the .lua file.
usage: python tools/gen_lua_cdef.py            print the cdef body
       python tools/gen_lua_cdef.py --write    rewrite the block between the BEGIN/END markers of tfluids_hip.lua"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tfluids_hip.h")
LUA = os.path.join(ROOT, "fluidnet_amd", "lua", "tfluids_hip.lua")
BEGIN, END = "-- BEGIN generated cdef (tools/gen_lua_cdef.py)", "-- END generated cdef"


def cdef_body():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    keep = []
    for line in src.splitlines():
        s = line.strip()
        if not s or s.startswith("#") or s == 'extern "C" {' or s == "}":
            continue
        keep.append(line.rstrip())
    body = "\n".join(keep)
    return re.sub(r"\n{2,}", "\n", body).strip() + "\n"


def main():
    body = cdef_body()
    if "--write" in sys.argv:
        lua = open(LUA).read()
        a, b = lua.index(BEGIN), lua.index(END)
        lua = lua[:a] + BEGIN + "\nffi.cdef[[\n" + body + "]]\n" + lua[b:]
        open(LUA, "w").write(lua)
    else:
        sys.stdout.write(body)


if __name__ == "__main__":
    main()
