// tuning.cpp -- the knob table of tuning.h read from the environment (once), and its description through the C ABI.
#include "tuning.h"
#include "../../include/kogarashi_amd.h"
#include <cstdlib>
#include <cstring>

namespace {
struct Row { const char* env; const char* doc; int dflt; int kg_tuning::*field; };
const Row ROWS[] = {
#define KG_X(field, env, def, doc) {env, doc, def, &kg_tuning::field},
    KG_TUNING_TABLE(KG_X)
#undef KG_X
};
constexpr int NROWS = (int)(sizeof(ROWS) / sizeof(ROWS[0]));

kg_tuning parse() {
  kg_tuning t;
  for (const Row& r : ROWS) {
    const char* e = getenv(r.env);
    if (!e || !*e) continue;
    if (r.field == &kg_tuning::msm_groups && strchr(e, ',')) { t.msm_groups_list = e; continue; }
    t.*(r.field) = atoi(e);
  }
  if (const char* e = getenv("KG_STREAM_PAD")) t.stream_pad = e;
  return t;
}
}  // namespace

namespace kg {
const kg_tuning& tuning() {
  static const kg_tuning t = parse();      // C++11 magic static: thread-safe, once
  return t;
}
}  // namespace kg

extern "C" int kg_tuning_describe(int index, const char** env, const char** doc, int* dflt, int* value) {
  if (index < 0) return NROWS;
  if (index >= NROWS) return KG_ERR_BAD_ARG;
  const Row& r = ROWS[index];
  if (env) *env = r.env;
  if (doc) *doc = r.doc;
  if (dflt) *dflt = r.dflt;
  if (value) *value = kg::tuning().*(r.field);
  return KG_OK;
}
