// Data formats either side of the proving path (host-only, no device work) -- SURVEY.md 8(a) rows a0', a0'', a18:
//   * the risc0 serde word stream of a String, as `ExecutorEnv::builder().write(&s)` feeds the guest its 13 inputs
//     (host/src/main.rs:389-417) and as `env::commit(&String)` leaves it in `receipt.journal.bytes`
//     (`[u32 LE length][utf8][zero padding to 4]`, host/src/main.rs:258-267; fixtures data/test/test.xml-Receipt-*.json);
//   * hyperfridge's own reading of the commitment: first '{' to last '}' of the journal (host/src/main.rs:258-267,
//     verifier/src/main.rs:176-185);
//   * the Receipt JSON envelope `serde_json::to_string(&receipt)` writes and `serde_json::from_slice` reads
//     (host/src/main.rs:251-252, verifier/src/main.rs:118-119): {"inner": ..., "journal": {"bytes": [...]}}.
// What the reference's fixtures pin: the envelope with "inner":"Fake" and the journal framing.  The composite layout
// (segments with seal / index / hashfn / verifier_parameters / claim, assumption_receipts, metadata) follows risc0-zkvm 3.x as
// recalled ("parity unpinned"); the digests inside it are computed in claim.cpp.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/r0hip.h"
#include "internal.hpp"
#include "receipt_types.hpp"

namespace {

// ---------------------------------------------------------------- a small JSON reader (objects, arrays, strings, integers)
struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj, NumArr } kind = Null;  // NumArr: an array of non-negative integers, kept flat (seals are long)
  bool b = false;
  double num = 0;
  bool integral = false;
  uint64_t u = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<uint64_t> nums;
  std::vector<std::pair<std::string, Json>> obj;
  const Json* get(const char* key) const {
    if (kind != Obj) return nullptr;
    for (const auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

struct Parser {
  const char* p;
  const char* end;
  std::string err;
  int depth = 0;
  bool fail(const char* what) {
    if (err.empty()) err = std::string(what) + " at byte " + std::to_string((size_t)(p - start));
    return false;
  }
  const char* start;
  void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
  bool lit(const char* s) {
    size_t n = strlen(s);
    if ((size_t)(end - p) < n || memcmp(p, s, n) != 0) return fail("unexpected token");
    p += n;
    return true;
  }
  bool string(std::string& out) {
    if (p >= end || *p != '"') return fail("expected a string");
    p++;
    while (p < end && *p != '"') {
      unsigned char c = (unsigned char)*p++;
      if (c < 0x20) return fail("control character in string");
      if (c != '\\') { out.push_back((char)c); continue; }
      if (p >= end) return fail("truncated escape");
      char e = *p++;
      switch (e) {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u': {
          if (end - p < 4) return fail("truncated \\u escape");
          unsigned v = 0;
          for (int i = 0; i < 4; i++) {
            char h = *p++;
            v = v * 16 + (h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : 256);
            if (v >= 0x10000) return fail("bad \\u escape");
          }
          if (v >= 0xD800 && v < 0xE000) return fail("surrogate escapes are not supported");
          if (v < 0x80) out.push_back((char)v);
          else if (v < 0x800) { out.push_back((char)(0xC0 | (v >> 6))); out.push_back((char)(0x80 | (v & 63))); }
          else { out.push_back((char)(0xE0 | (v >> 12))); out.push_back((char)(0x80 | ((v >> 6) & 63))); out.push_back((char)(0x80 | (v & 63))); }
          break;
        }
        default: return fail("unknown escape");
      }
    }
    if (p >= end) return fail("unterminated string");
    p++;
    return true;
  }
  bool value(Json& out) {
    if (++depth > 64) return fail("nesting too deep");
    ws();
    if (p >= end) return fail("unexpected end");
    bool ok = true;
    if (*p == '{') {
      out.kind = Json::Obj;
      p++;
      ws();
      if (p < end && *p == '}') { p++; depth--; return true; }
      while (ok) {
        ws();
        std::string key;
        if (!string(key)) return false;
        ws();
        if (p >= end || *p != ':') return fail("expected ':'");
        p++;
        out.obj.emplace_back(key, Json());
        if (!value(out.obj.back().second)) return false;
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == '}') { p++; break; }
        return fail("expected ',' or '}'");
      }
    } else if (*p == '[') {
      out.kind = Json::Arr;
      p++;
      ws();
      if (p < end && *p == ']') { p++; depth--; return true; }
      if (p < end && *p >= '0' && *p <= '9') {  // flat path for arrays of non-negative integers (seal words, journal bytes)
        const char* save = p;
        bool flat = true;
        while (true) {
          ws();
          if (p >= end || *p < '0' || *p > '9') { flat = false; break; }
          uint64_t u = 0;
          bool overflow = false;
          while (p < end && *p >= '0' && *p <= '9') {
            if (u > (UINT64_MAX - 9) / 10) overflow = true;
            u = u * 10 + (uint64_t)(*p++ - '0');
          }
          if (overflow || (p < end && (*p == '.' || *p == 'e' || *p == 'E'))) { flat = false; break; }
          out.nums.push_back(u);
          ws();
          if (p < end && *p == ',') { p++; continue; }
          if (p < end && *p == ']') { p++; break; }
          flat = false;
          break;
        }
        if (flat) { out.kind = Json::NumArr; depth--; return true; }
        p = save;  // something else in there: parse it the general way
        out.nums.clear();
      }
      while (ok) {
        out.arr.emplace_back();
        if (!value(out.arr.back())) return false;
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == ']') { p++; break; }
        return fail("expected ',' or ']'");
      }
    } else if (*p == '"') {
      out.kind = Json::Str;
      ok = string(out.str);
    } else if (*p == 't') { out.kind = Json::Bool; out.b = true; ok = lit("true"); }
    else if (*p == 'f') { out.kind = Json::Bool; out.b = false; ok = lit("false"); }
    else if (*p == 'n') { out.kind = Json::Null; ok = lit("null"); }
    else {
      const char* s = p;
      bool neg = p < end && *p == '-';
      if (neg) p++;
      if (p >= end || *p < '0' || *p > '9') return fail("unexpected character");
      uint64_t u = 0;
      bool integral = !neg, overflow = false;
      while (p < end && *p >= '0' && *p <= '9') {
        if (u > (UINT64_MAX - 9) / 10) overflow = true;
        u = u * 10 + (uint64_t)(*p++ - '0');
      }
      if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) {
        integral = false;
        while (p < end && (*p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-' || (*p >= '0' && *p <= '9'))) p++;
      }
      out.kind = Json::Num;
      out.integral = integral && !overflow;
      out.u = u;
      out.num = strtod(std::string(s, p).c_str(), nullptr);
    }
    depth--;
    return ok;
  }
};

}  // namespace

struct r0h_env {
  std::vector<uint32_t> words;
};

namespace {

const char* u32_array(const Json* j, uint64_t limit, const char* what, std::vector<uint64_t>& out) {
  R0H_REQUIRE(j && (j->kind == Json::Arr || j->kind == Json::NumArr), "receipt JSON: %s is not an array", what);
  if (j->kind == Json::NumArr) {
    for (uint64_t u : j->nums)
      R0H_REQUIRE(u <= limit, "receipt JSON: %s holds something other than an integer in [0, %llu]", what, (unsigned long long)limit);
    out = j->nums;
    return nullptr;
  }
  out.reserve(j->arr.size());
  for (const Json& v : j->arr) {
    R0H_REQUIRE(v.kind == Json::Num && v.integral && v.u <= limit, "receipt JSON: %s holds something other than an integer in [0, %llu]", what,
                (unsigned long long)limit);
    out.push_back(v.u);
  }
  return nullptr;
}

// a risc0 `Digest` in serde_json: 64 hex digits (human-readable form); the eight-u32 array form of the binary codecs is accepted too
const char* parse_digest(const Json* j, const char* what, uint8_t out[32]) {
  R0H_REQUIRE(j, "receipt JSON: no %s", what);
  if (j->kind == Json::Str) {
    R0H_REQUIRE(j->str.size() == 64, "receipt JSON: %s is not 64 hex digits", what);
    for (int i = 0; i < 32; i++) {
      unsigned v = 0;
      for (int k = 0; k < 2; k++) {
        const char h = j->str[2 * i + k];
        const unsigned d = h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : 256;
        R0H_REQUIRE(d < 16, "receipt JSON: %s is not 64 hex digits", what);
        v = v * 16 + d;
      }
      out[i] = (uint8_t)v;
    }
    return nullptr;
  }
  std::vector<uint64_t> w;
  R0H_TRY(u32_array(j, 0xffffffffull, what, w));
  R0H_REQUIRE(w.size() == 8, "receipt JSON: %s is not a digest (8 words)", what);
  for (int i = 0; i < 8; i++)
    for (int b = 0; b < 4; b++) out[4 * i + b] = (uint8_t)(w[i] >> (8 * b));
  return nullptr;
}

// MaybePruned<T>: {"Value": T} or {"Pruned": digest}
const Json* maybe_pruned(const Json* j, bool* pruned) {
  if (!j || j->kind != Json::Obj || j->obj.size() != 1) return nullptr;
  if (j->obj[0].first == "Value") { *pruned = false; return &j->obj[0].second; }
  if (j->obj[0].first == "Pruned") { *pruned = true; return &j->obj[0].second; }
  return nullptr;
}

const char* parse_system_state(const Json* j, const char* what, r0h_system_state& st) {
  bool pruned = false;
  const Json* v = maybe_pruned(j, &pruned);
  R0H_REQUIRE(v && !pruned && v->kind == Json::Obj, "receipt JSON: claim.%s is not {\"Value\":{pc, merkle_root}} (a pruned system state cannot be chained)", what);
  const Json* pc = v->get("pc");
  R0H_REQUIRE(pc && pc->kind == Json::Num && pc->integral && pc->u <= 0xffffffffull, "receipt JSON: claim.%s.pc is not a u32", what);
  st.pc = (uint32_t)pc->u;
  return parse_digest(v->get("merkle_root"), "claim system state merkle_root", st.merkle_root);
}

const char* parse_claim(const Json& c, r0h_receipt_claim& out, const std::vector<uint8_t>&) {
  R0H_REQUIRE(c.kind == Json::Obj, "receipt JSON: a claim is not an object");
  memset(&out, 0, sizeof out);
  R0H_TRY(parse_system_state(c.get("pre"), "pre", out.pre));
  R0H_TRY(parse_system_state(c.get("post"), "post", out.post));
  const Json* ec = c.get("exit_code");
  R0H_REQUIRE(ec, "receipt JSON: claim without exit_code");
  if (ec->kind == Json::Str) {
    R0H_REQUIRE(ec->str == "SystemSplit" || ec->str == "SessionLimit", "receipt JSON: unknown exit code \"%s\"", ec->str.c_str());
    out.exit_system = 2;
    out.exit_user = ec->str == "SessionLimit" ? 2 : 0;
  } else {
    R0H_REQUIRE(ec->kind == Json::Obj && ec->obj.size() == 1 && ec->obj[0].second.kind == Json::Num && ec->obj[0].second.integral && ec->obj[0].second.u <= 0xffffffffull,
                "receipt JSON: exit_code is neither a unit variant nor {\"Halted\"|\"Paused\": u32}");
    R0H_REQUIRE(ec->obj[0].first == "Halted" || ec->obj[0].first == "Paused", "receipt JSON: unknown exit code \"%s\"", ec->obj[0].first.c_str());
    out.exit_system = ec->obj[0].first == "Paused" ? 1 : 0;
    out.exit_user = (uint32_t)ec->obj[0].second.u;
  }
  bool pruned = false;
  const Json* in = maybe_pruned(c.get("input"), &pruned);
  R0H_REQUIRE(in, "receipt JSON: claim.input is not a MaybePruned value");
  if (pruned) R0H_TRY(parse_digest(in, "claim.input", out.input_digest));
  else R0H_REQUIRE(in->kind == Json::Null, "receipt JSON: a claim with an unpruned input is not supported");
  const Json* o = maybe_pruned(c.get("output"), &pruned);
  R0H_REQUIRE(o, "receipt JSON: claim.output is not a MaybePruned value");
  if (pruned) {
    R0H_TRY(parse_digest(o, "claim.output", out.output_digest));
  } else if (o->kind != Json::Null) {  // {"Value":{"journal":{..},"assumptions":{..}}}: fold it to its digest
    R0H_REQUIRE(o->kind == Json::Obj, "receipt JSON: claim.output value is not an object");
    uint8_t jd[32], ad[32] = {0};
    bool jp = false, ap = false;
    const Json* jv = maybe_pruned(o->get("journal"), &jp);
    R0H_REQUIRE(jv, "receipt JSON: claim.output.journal is not a MaybePruned value");
    if (jp) {
      R0H_TRY(parse_digest(jv, "claim.output.journal", jd));
    } else {
      std::vector<uint64_t> bytes;
      R0H_TRY(u32_array(jv, 255, "claim.output.journal", bytes));
      std::vector<uint8_t> raw(bytes.begin(), bytes.end());
      R0H_TRY(r0h_sha256(raw.data(), raw.size(), jd));
    }
    const Json* av = maybe_pruned(o->get("assumptions"), &ap);
    R0H_REQUIRE(av, "receipt JSON: claim.output.assumptions is not a MaybePruned value");
    if (ap) R0H_TRY(parse_digest(av, "claim.output.assumptions", ad));
    else R0H_REQUIRE((av->kind == Json::Arr && av->arr.empty()) || (av->kind == Json::NumArr && av->nums.empty()), "receipt JSON: unresolved assumptions are not supported");
    uint8_t down[2][32];
    memcpy(down[0], jd, 32);
    memcpy(down[1], ad, 32);
    r0h::tagged_struct("risc0.Output", down, 2, nullptr, 0, out.output_digest);
  }
  return nullptr;
}

const char* parse_receipt(const Json& root, r0h_receipt& rc) {
  R0H_REQUIRE(root.kind == Json::Obj, "receipt JSON: top level is not an object");
  const Json* journal = root.get("journal");
  R0H_REQUIRE(journal && journal->kind == Json::Obj, "receipt JSON: no \"journal\" object");
  std::vector<uint64_t> bytes;
  R0H_TRY(u32_array(journal->get("bytes"), 255, "journal.bytes", bytes));
  rc.journal.assign(bytes.begin(), bytes.end());
  const Json* inner = root.get("inner");
  R0H_REQUIRE(inner, "receipt JSON: no \"inner\"");
  if (inner->kind == Json::Str) {  // unit variant, risc0 0.19 style: "inner":"Fake" (what the reference's fixtures hold; no metadata)
    R0H_REQUIRE(inner->str == "Fake", "receipt JSON: unsupported inner receipt \"%s\"", inner->str.c_str());
    rc.kind = R0H_RECEIPT_FAKE;
    return nullptr;
  }
  R0H_REQUIRE(inner->kind == Json::Obj && inner->obj.size() == 1, "receipt JSON: \"inner\" is neither a variant name nor a one-key object");
  const std::string& variant = inner->obj[0].first;
  const Json& body = inner->obj[0].second;
  if (variant == "Fake") { rc.kind = R0H_RECEIPT_FAKE; return nullptr; }
  R0H_REQUIRE(variant == "Composite", "receipt JSON: unsupported inner receipt \"%s\"", variant.c_str());
  rc.kind = R0H_RECEIPT_COMPOSITE;
  const Json* segs = body.get("segments");
  R0H_REQUIRE(segs && segs->kind == Json::Arr, "receipt JSON: Composite without \"segments\"");
  for (const Json& s : segs->arr) {
    R0H_REQUIRE(s.kind == Json::Obj, "receipt JSON: a segment is not an object");
    r0h_receipt::Segment seg;
    std::vector<uint64_t> words;
    R0H_TRY(u32_array(s.get("seal"), 0xffffffffull, "segment.seal", words));
    seg.seal.assign(words.begin(), words.end());
    const Json* idx = s.get("index");
    R0H_REQUIRE(idx && idx->kind == Json::Num && idx->integral && idx->u <= 0xffffffffull, "receipt JSON: segment without an integer \"index\"");
    seg.index = (uint32_t)idx->u;
    const Json* hf = s.get("hashfn");
    R0H_REQUIRE(hf && hf->kind == Json::Str, "receipt JSON: segment without \"hashfn\"");
    seg.hashfn = hf->str;
    if (const Json* vp = s.get("verifier_parameters")) R0H_TRY(parse_digest(vp, "segment.verifier_parameters", seg.verifier_parameters));
    const Json* claim = s.get("claim");
    if (claim && claim->kind != Json::Null) {
      R0H_TRY(parse_claim(*claim, seg.claim, rc.journal));
      seg.has_claim = true;
    }
    rc.segments.push_back(std::move(seg));
  }
  if (const Json* ip = body.get("image_proof")) {  // this library's own addition: absent from risc0's receipts
    R0H_REQUIRE(ip->kind == Json::Obj, "receipt JSON: \"image_proof\" is not an object");
    std::vector<uint64_t> words;
    R0H_TRY(u32_array(ip->get("seal"), 0xffffffffull, "image_proof.seal", words));
    rc.image_seal.assign(words.begin(), words.end());
  }
  if (const Json* md = root.get("metadata")) {
    R0H_REQUIRE(md->kind == Json::Obj, "receipt JSON: \"metadata\" is not an object");
    R0H_TRY(parse_digest(md->get("verifier_parameters"), "metadata.verifier_parameters", rc.verifier_parameters));
    rc.has_metadata = true;
  }
  return nullptr;
}

void append_hex(std::string& s, const uint8_t d[32]) {
  static const char hex[] = "0123456789abcdef";
  s += '"';
  for (int i = 0; i < 32; i++) { s += hex[d[i] >> 4]; s += hex[d[i] & 15]; }
  s += '"';
}

void append_escaped(std::string& s, const std::string& v) {
  for (unsigned char c : v) {
    if (c == '"' || c == '\\') { s += '\\'; s += (char)c; }
    else if (c < 0x20) { char tmp[8]; snprintf(tmp, sizeof tmp, "\\u%04x", c); s += tmp; }
    else s += (char)c;
  }
}

void append_u(std::string& s, uint64_t v) {
  char tmp[24];
  snprintf(tmp, sizeof tmp, "%llu", (unsigned long long)v);
  s += tmp;
}

}  // namespace

extern "C" {

const char* r0h_serde_encode_str(const uint8_t* utf8, size_t len, uint8_t* out, size_t capacity, size_t* out_len) {
  R0H_REQUIRE((utf8 || len == 0) && out_len, "r0h_serde_encode_str: NULL argument");
  R0H_REQUIRE(len <= 0xffffffffull, "r0h_serde_encode_str: a serde length prefix is 32 bits");
  const size_t total = 4 + ((len + 3) & ~(size_t)3);
  *out_len = total;
  if (!out) return nullptr;  // size query
  R0H_REQUIRE(capacity >= total, "r0h_serde_encode_str: %zu bytes needed, capacity is %zu", total, capacity);
  out[0] = (uint8_t)len; out[1] = (uint8_t)(len >> 8); out[2] = (uint8_t)(len >> 16); out[3] = (uint8_t)(len >> 24);
  if (len) memcpy(out + 4, utf8, len);
  memset(out + 4 + len, 0, total - 4 - len);
  return nullptr;
}

const char* r0h_serde_decode_str(const uint8_t* bytes, size_t n, size_t* str_off, size_t* str_len, size_t* consumed) {
  R0H_REQUIRE(bytes && str_off && str_len, "r0h_serde_decode_str: NULL argument");
  R0H_REQUIRE(n >= 4, "serde string: %zu bytes cannot hold the length word", n);
  const size_t len = (size_t)bytes[0] | ((size_t)bytes[1] << 8) | ((size_t)bytes[2] << 16) | ((size_t)bytes[3] << 24);
  const size_t total = 4 + ((len + 3) & ~(size_t)3);
  R0H_REQUIRE(total <= n, "serde string: length word says %zu bytes, only %zu follow", len, n - 4);
  for (size_t i = 4 + len; i < total; i++) R0H_REQUIRE(bytes[i] == 0, "serde string: non-zero padding byte at offset %zu", i);
  *str_off = 4;
  *str_len = len;
  if (consumed) *consumed = total;
  return nullptr;
}

// ---- the ExecutorEnv input stream: host/src/main.rs:389-417 writes 12 Strings and one Vec<u8> (the decrypted transaction key),
// the guest reads them back in the same order (methods/guest/src/main.rs:159-171).  A String is the frame above; a Vec<u8>
// without serde_bytes goes element by element, one u32 word per byte (risc0 serde `serialize_u8`, as recalled: unpinned).
const char* r0h_env_new(r0h_env** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(out, "r0h_env_new: NULL argument");
  *out = new r0h_env;
  return nullptr;
  R0H_GUARD_END
}
const char* r0h_env_write_str(r0h_env* e, const uint8_t* utf8, size_t len) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(e && (utf8 || len == 0), "r0h_env_write_str: NULL argument");
  R0H_REQUIRE(len <= 0xffffffffull, "r0h_env_write_str: a serde length prefix is 32 bits");
  e->words.push_back((uint32_t)len);
  for (size_t i = 0; i < len; i += 4) {
    uint32_t w = 0;
    for (size_t b = 0; b < 4 && i + b < len; b++) w |= (uint32_t)utf8[i + b] << (8 * b);
    e->words.push_back(w);
  }
  return nullptr;
  R0H_GUARD_END
}
const char* r0h_env_write_u8_seq(r0h_env* e, const uint8_t* bytes, size_t len) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(e && (bytes || len == 0), "r0h_env_write_u8_seq: NULL argument");
  R0H_REQUIRE(len <= 0xffffffffull, "r0h_env_write_u8_seq: a serde length prefix is 32 bits");
  e->words.push_back((uint32_t)len);
  for (size_t i = 0; i < len; i++) e->words.push_back(bytes[i]);
  return nullptr;
  R0H_GUARD_END
}
const char* r0h_env_words(const r0h_env* e, const uint32_t** words, size_t* n_words) {
  R0H_REQUIRE(e && words && n_words, "r0h_env_words: NULL argument");
  *words = e->words.data();
  *n_words = e->words.size();
  return nullptr;
}
const char* r0h_env_free(r0h_env* e) {
  delete e;
  return nullptr;
}

const char* r0h_journal_commitment_span(const uint8_t* bytes, size_t n, size_t* off, size_t* len) {
  R0H_REQUIRE((bytes || n == 0) && off && len, "r0h_journal_commitment_span: NULL argument");
  size_t first = 0, last = n;  // the reference's defaults when a brace is missing: position(..).unwrap_or(0), rposition(..).unwrap_or(len)
  for (size_t i = 0; i < n; i++)
    if (bytes[i] == '{') { first = i; break; }
  for (size_t i = n; i-- > 0;)
    if (bytes[i] == '}') { last = i; break; }
  R0H_REQUIRE(last < n && first <= last, "journal: no JSON object between the first '{' and the last '}'");
  *off = first;
  *len = last - first + 1;
  return nullptr;
}

const char* r0h_receipt_parse(const char* json, size_t n, r0h_receipt** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(json && out, "r0h_receipt_parse: NULL argument");
  Parser ps{json, json + n, std::string(), 0, json};
  Json root;
  bool ok = ps.value(root);
  if (ok) { ps.ws(); if (ps.p != ps.end) ok = ps.fail("trailing characters"); }
  R0H_REQUIRE(ok, "receipt JSON: %s", ps.err.c_str());
  std::unique_ptr<r0h_receipt> rc(new r0h_receipt);
  R0H_TRY(parse_receipt(root, *rc));
  *out = rc.release();
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_receipt_new(int kind, const uint8_t* journal, size_t journal_len, r0h_receipt** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(out && (journal || journal_len == 0), "r0h_receipt_new: NULL argument");
  R0H_REQUIRE(kind == R0H_RECEIPT_FAKE || kind == R0H_RECEIPT_COMPOSITE, "r0h_receipt_new: unknown kind %d", kind);
  r0h_receipt* rc = new r0h_receipt;
  rc->kind = kind;
  rc->journal.assign(journal, journal + journal_len);
  rc->has_metadata = kind == R0H_RECEIPT_COMPOSITE;  // what this library writes carries risc0 3.x's metadata; Fake keeps the fixtures' form
  *out = rc;
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_receipt_add_segment(r0h_receipt* rc, const uint32_t* seal, size_t seal_words, uint32_t index) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(rc && seal, "r0h_receipt_add_segment: NULL argument");
  R0H_REQUIRE(rc->kind == R0H_RECEIPT_COMPOSITE, "r0h_receipt_add_segment: a Fake receipt has no segments");
  r0h_receipt::Segment s;
  s.seal.assign(seal, seal + seal_words);
  s.index = index;
  s.hashfn = "poseidon2";
  rc->segments.push_back(std::move(s));
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_receipt_add_segment_claim(r0h_receipt* rc, const uint32_t* seal, size_t seal_words, uint32_t index, const r0h_receipt_claim* claim,
                                          const uint8_t* verifier_parameters) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(claim, "r0h_receipt_add_segment_claim: NULL claim");
  R0H_REQUIRE(claim->exit_system <= 2 && (claim->exit_system != 2 || claim->exit_user == 0 || claim->exit_user == 2), "r0h_receipt_add_segment_claim: exit code (%u, %u) names no ExitCode variant",
              claim->exit_system, claim->exit_user);
  R0H_TRY(r0h_receipt_add_segment(rc, seal, seal_words, index));
  r0h_receipt::Segment& s = rc->segments.back();
  s.has_claim = true;
  s.claim = *claim;
  if (verifier_parameters) memcpy(s.verifier_parameters, verifier_parameters, 32);
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_receipt_merge(const r0h_receipt* const* parts, size_t n, r0h_receipt** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(parts && n && out, "r0h_receipt_merge: NULL argument");
  size_t total = 0;
  for (size_t k = 0; k < n; k++) {
    R0H_REQUIRE(parts[k] && parts[k]->kind == R0H_RECEIPT_COMPOSITE, "r0h_receipt_merge: receipt %zu is not a composite receipt", k);
    R0H_REQUIRE(parts[k]->journal == parts[0]->journal, "r0h_receipt_merge: receipt %zu carries another journal", k);
    total += parts[k]->segments.size();
  }
  std::vector<const r0h_receipt::Segment*> at(total, nullptr);
  for (size_t k = 0; k < n; k++)
    for (const r0h_receipt::Segment& s : parts[k]->segments) {
      R0H_REQUIRE(s.index < total, "r0h_receipt_merge: segment index %u with %zu segments in all: one is missing", s.index, total);
      R0H_REQUIRE(!at[s.index], "r0h_receipt_merge: segment %u is there twice", s.index);
      at[s.index] = &s;
    }
  std::unique_ptr<r0h_receipt> rc(new r0h_receipt(*parts[0]));
  rc->segments.clear();
  for (size_t i = 0; i < total; i++) rc->segments.push_back(*at[i]);  // (every slot is filled: total indices below total, none twice)
  for (size_t k = 0; k < n; k++) {  // the session's image proof travels with whichever part made it
    if (parts[k]->image_seal.empty()) continue;
    R0H_REQUIRE(rc->image_seal.empty() || rc->image_seal == parts[k]->image_seal, "r0h_receipt_merge: receipt %zu carries another image proof", k);
    rc->image_seal = parts[k]->image_seal;
  }
  *out = rc.release();
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_receipt_segment_claim(const r0h_receipt* rc, size_t i, r0h_receipt_claim* claim_out, int* has_claim_out) {
  R0H_REQUIRE(rc && claim_out && has_claim_out, "r0h_receipt_segment_claim: NULL argument");
  R0H_REQUIRE(i < rc->segments.size(), "r0h_receipt_segment_claim: segment %zu of %zu", i, rc->segments.size());
  *has_claim_out = rc->segments[i].has_claim ? 1 : 0;
  if (rc->segments[i].has_claim) *claim_out = rc->segments[i].claim;
  return nullptr;
}

const char* r0h_receipt_set_image_proof(r0h_receipt* rc, const uint32_t* seal, size_t seal_words) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(rc && (seal || !seal_words), "r0h_receipt_set_image_proof: NULL argument");
  R0H_REQUIRE(rc->kind == R0H_RECEIPT_COMPOSITE, "r0h_receipt_set_image_proof: not a composite receipt");
  rc->image_seal.assign(seal, seal + seal_words);
  return nullptr;
  R0H_GUARD_END
}
const char* r0h_receipt_image_proof(const r0h_receipt* rc, const uint32_t** seal, size_t* seal_words) {
  R0H_REQUIRE(rc && seal && seal_words, "r0h_receipt_image_proof: NULL argument");
  *seal = rc->image_seal.empty() ? nullptr : rc->image_seal.data();
  *seal_words = rc->image_seal.size();
  return nullptr;
}

const char* r0h_receipt_free(r0h_receipt* rc) {
  delete rc;
  return nullptr;
}

int r0h_receipt_kind(const r0h_receipt* rc) { return rc ? rc->kind : -1; }
size_t r0h_receipt_n_segments(const r0h_receipt* rc) { return rc ? rc->segments.size() : 0; }

const char* r0h_receipt_journal(const r0h_receipt* rc, const uint8_t** bytes, size_t* n) {
  R0H_REQUIRE(rc && bytes && n, "r0h_receipt_journal: NULL argument");
  *bytes = rc->journal.data();
  *n = rc->journal.size();
  return nullptr;
}

const char* r0h_receipt_segment(const r0h_receipt* rc, size_t i, const uint32_t** seal, size_t* seal_words, uint32_t* index) {
  R0H_REQUIRE(rc && seal && seal_words, "r0h_receipt_segment: NULL argument");
  R0H_REQUIRE(i < rc->segments.size(), "r0h_receipt_segment: segment %zu of %zu", i, rc->segments.size());
  *seal = rc->segments[i].seal.data();
  *seal_words = rc->segments[i].seal.size();
  if (index) *index = rc->segments[i].index;
  return nullptr;
}

// serde_json's compact form (no spaces, keys in struct order): byte-identical to the reference's fixtures for Fake receipts
const char* r0h_receipt_to_json(const r0h_receipt* rc, char** json_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(rc && json_out, "r0h_receipt_to_json: NULL argument");
  std::string s = "{\"inner\":";
  if (rc->kind == R0H_RECEIPT_FAKE) {
    s += "\"Fake\"";
  } else {
    s += "{\"Composite\":{\"segments\":[";
    for (size_t i = 0; i < rc->segments.size(); i++) {
      const r0h_receipt::Segment& g = rc->segments[i];
      if (i) s += ',';
      s += "{\"seal\":[";
      for (size_t w = 0; w < g.seal.size(); w++) { if (w) s += ','; append_u(s, g.seal[w]); }
      s += "],\"index\":";
      append_u(s, g.index);
      s += ",\"hashfn\":\"";
      append_escaped(s, g.hashfn);
      s += "\",\"verifier_parameters\":";
      append_hex(s, g.verifier_parameters);
      s += ",\"claim\":";
      if (!g.has_claim) {
        s += "null";
      } else {
        const r0h_receipt_claim& c = g.claim;
        static const uint8_t zero[32] = {0};
        auto state = [&](const char* key, const r0h_system_state& st) {
          s += '"'; s += key; s += "\":{\"Value\":{\"pc\":";
          append_u(s, st.pc);
          s += ",\"merkle_root\":";
          append_hex(s, st.merkle_root);
          s += "}}";
        };
        s += '{';
        state("pre", c.pre);
        s += ',';
        state("post", c.post);
        s += ",\"exit_code\":";
        if (c.exit_system == 2) s += c.exit_user == 2 ? "\"SessionLimit\"" : "\"SystemSplit\"";
        else { s += c.exit_system == 1 ? "{\"Paused\":" : "{\"Halted\":"; append_u(s, c.exit_user); s += '}'; }
        s += ",\"input\":{\"Pruned\":";
        append_hex(s, c.input_digest);
        s += "},\"output\":";
        if (memcmp(c.output_digest, zero, 32) == 0) s += "{\"Value\":null}";
        else { s += "{\"Pruned\":"; append_hex(s, c.output_digest); s += '}'; }
        s += '}';
      }
      s += '}';
    }
    s += "],\"assumption_receipts\":[],\"verifier_parameters\":";
    append_hex(s, rc->verifier_parameters);
    if (!rc->image_seal.empty()) {
      s += ",\"image_proof\":{\"seal\":[";
      for (size_t w = 0; w < rc->image_seal.size(); w++) { if (w) s += ','; append_u(s, rc->image_seal[w]); }
      s += "]}";
    }
    s += "}}";
  }
  s += ",\"journal\":{\"bytes\":[";
  for (size_t i = 0; i < rc->journal.size(); i++) { if (i) s += ','; append_u(s, rc->journal[i]); }
  s += "]}";
  if (rc->has_metadata) { s += ",\"metadata\":{\"verifier_parameters\":"; append_hex(s, rc->verifier_parameters); s += '}'; }
  s += '}';
  char* out = (char*)malloc(s.size() + 1);
  R0H_REQUIRE(out, "r0h_receipt_to_json: out of memory");
  memcpy(out, s.c_str(), s.size() + 1);
  *json_out = out;
  return nullptr;
  R0H_GUARD_END
}

}  // extern "C"
