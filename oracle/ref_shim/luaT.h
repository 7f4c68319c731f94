// Minimal stand-in for lua.h/luaT.h -- TEST INFRASTRUCTURE ONLY (see TH.h in this directory).
//
// The reference's native entry points are `static int tfluids_<Real>Main_<op>(lua_State*)` and
// read their arguments positionally off the Lua stack. Here the "stack" is a plain array that
// the C trampolines in ../ref_wrap.cc fill in; return values pushed by lua_pushnumber are
// captured into `ret`.
#pragma once
#include <string>
#include <vector>
#include "TH.h"

struct ShimArg {
  double num = 0.0;
  void* ptr = nullptr;
  const char* str = nullptr;
  bool is_bool = false;
};

struct lua_State {
  std::vector<ShimArg> a;   // 1-based positions map to a[pos-1]
  std::vector<double> ret;
};

inline const ShimArg& shim_arg(lua_State* L, int i) {
  if (i < 1 || i > (int)L->a.size()) throw ShimError("shim: lua arg index out of range");
  return L->a[i - 1];
}
inline double lua_tonumber(lua_State* L, int i) { return shim_arg(L, i).num; }
inline long lua_tointeger(lua_State* L, int i) { return (long)shim_arg(L, i).num; }
inline long luaL_checkinteger(lua_State* L, int i) { return (long)shim_arg(L, i).num; }
inline int lua_toboolean(lua_State* L, int i) { return shim_arg(L, i).num != 0.0; }
inline int lua_isboolean(lua_State* L, int i) { return shim_arg(L, i).is_bool; }
inline const char* lua_tostring(lua_State* L, int i) { return shim_arg(L, i).str; }
inline void* luaT_checkudata(lua_State* L, int i, const char*) {
  void* p = shim_arg(L, i).ptr;
  if (!p) throw ShimError("shim: expected a tensor argument");
  return p;
}
[[noreturn]] inline int luaL_error(lua_State*, const char* fmt, ...) { throw ShimError(fmt); }
inline void lua_pushnumber(lua_State* L, double v) { L->ret.push_back(v); }
inline void lua_pushboolean(lua_State*, int) {}
inline void lua_pushstring(lua_State*, const char*) {}
inline void lua_pushvalue(lua_State*, int) {}
inline void lua_newtable(lua_State*) {}
inline void lua_settable(lua_State*, int) {}
inline void lua_setfield(lua_State*, int, const char*) {}
inline void lua_setglobal(lua_State*, const char*) {}

typedef int (*lua_CFunction)(lua_State*);
struct luaL_Reg { const char* name; lua_CFunction func; };
inline void luaT_setfuncs(lua_State*, const luaL_Reg*, int) {}
inline void luaT_pushmetatable(lua_State*, const char*) {}
inline void luaT_registeratname(lua_State*, const luaL_Reg*, const char*) {}

#define LUA_EXTERNC extern "C"
#define DLL_EXPORT
