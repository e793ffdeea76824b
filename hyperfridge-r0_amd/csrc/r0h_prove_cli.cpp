// r0h_prove: a minimal compiled host over the C ABI (no Python, no torch): load a circuit blob, generate a synthetic
// witness on the device, prove K segments, print throughput and a digest of the seal.
// It stands where hyperfridge's `host prove-camt53` stands relative to risc0 (host/src/main.rs:420-423 obtains a prover and
// calls prove once); everything risc0-specific above the segment prover (executor, receipts) is out of scope.
//   usage: r0h_prove <circuit.r0c> [--code-object file.hsaco] [--po2 N] [--segments K] [--seed S] [--device D] [--contexts C] [--seal-out file] [--verify 1]
//                    [--receipt-out file.json | --receipt-dir dir] [--journal text] [--receipts R]
// With --receipt-out / --receipt-dir every segment is proved for a claim (risc0-zkvm `ReceiptClaim`): the session's system states
// are synthetic names (there is no executor here), segment k runs from state k to state k+1, all but the last end in SystemSplit,
// the last halts with the journal's output; the claim's eight naming words are planted as the segment's public inputs.  The image
// id (digest of state 0) and the control root of the trace size are printed for the verifier (`r0h_verify --receipt`).
//
//   usage: r0h_prove <trace.r0c> --elf guest.elf --input words.bin [--code-object file.hsaco] [--po2 N] [--device D] --receipt-out file.json
//                    [--receipts R --contexts C --receipt-dir dir]   R sessions of the guest, C in flight (BASELINE.json configs[3]), each receipt verified with the ELF
//                    [--image-circuit circuits/image.r0c --image-code-object circuits/image.evalcheck.hsaco]   the receipts carry an image proof and are also
//                    verified with the image id alone (r0h_receipt_verify_image: the reference's `receipt.verify(image_id)`)
// `prover.prove(env, ELF)` itself (host/src/main.rs:420-423) as a compiled host: the guest ELF is executed on the u32 input stream
// (the ExecutorEnv frames, little-endian words in a file), every segment is proved (r0h_prove_elf: with circuits/trace.r0c the
// seals are proofs over the segments' own cycles), the receipt is written as JSON, and one line of JSON names the image id in
// the reference's IMAGE_ID.hex form, the control root of every trace size used, the cycle and segment counts and the timing.
//
//   usage: r0h_prove <trace.r0c> --elf circuits/guest_camt53.elf --camt53-response response.xml --pub-bank b.pem --pub-client c.pem --pub-witness w.pem
//                    --tx-key-raw TransactionKeyDecrypt.bin --witness-hex Witness.hex --iban CH.. [--hostinfo text] [--form 1] --receipt-out file.json
// the reference's `host` from its inputs on (host/src/main.rs:389-423): instead of --input, the guest's input words are built from the
// EBICS response, the three public keys, the decrypted transaction key and the witness signature (r0h_camt53_guest_input; r0h_preprocess
// makes the last two from the private keys) -- response.xml to receipt with the compiled hosts alone.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/r0hip.h"

static void die(const char* what, const char* err) {
  fprintf(stderr, "r0h_prove: %s: %s\n", what, err);
  r0h_free_error(err);
  exit(2);
}
#define CHECK(call)                      \
  do {                                   \
    const char* e__ = (call);            \
    if (e__) die(#call, e__);            \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2 || !strcmp(argv[1], "--help") || !strcmp(argv[1], "-h")) {
    printf("usage: r0h_prove <circuit.r0c> [--code-object file.hsaco] [--po2 N] [--segments K] [--seed S] [--device D] [--contexts C] [--seal-out file] [--verify 1] [--receipt-out file.json | --receipt-dir dir] [--journal text] [--receipts R]\n"
           "       r0h_prove <trace.r0c> --elf guest.elf --input words.bin [--code-object file.hsaco] [--po2 N] [--device D] --receipt-out file.json\n%s\n", r0h_version());
    return argc < 2 ? 1 : 0;
  }
  std::string blob_path = argv[1], co_path, seal_out, receipt_out, receipt_dir, journal_text, elf_path, input_path;
  std::map<std::string, std::string> camt;  // --camt53-response and what goes with it
  std::string receipt_prefix;               // --receipt-prefix P: the reference's file name, P-Receipt-<image id>-latest.json (host/src/main.rs:312-316)
  std::string image_circuit_path, image_co_path;
  unsigned po2 = 16, segments = 1, device = 0, contexts = 1, verify = 0, receipts = 1;
  unsigned long long seed = 1;
  for (int i = 2; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--code-object")) co_path = argv[i + 1];
    else if (!strcmp(argv[i], "--po2")) po2 = (unsigned)atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--segments")) segments = (unsigned)atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--seed")) seed = strtoull(argv[i + 1], nullptr, 10);
    else if (!strcmp(argv[i], "--device")) device = (unsigned)atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--contexts")) contexts = (unsigned)atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--seal-out")) seal_out = argv[i + 1];
    else if (!strcmp(argv[i], "--verify")) verify = (unsigned)atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--receipt-out")) receipt_out = argv[i + 1];
    else if (!strcmp(argv[i], "--receipt-dir")) receipt_dir = argv[i + 1];
    else if (!strcmp(argv[i], "--journal")) journal_text = argv[i + 1];
    else if (!strcmp(argv[i], "--receipts")) receipts = (unsigned)atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--elf")) elf_path = argv[i + 1];
    else if (!strcmp(argv[i], "--input")) input_path = argv[i + 1];
    else if (!strcmp(argv[i], "--receipt-prefix")) receipt_prefix = argv[i + 1];
    else if (!strcmp(argv[i], "--image-circuit")) image_circuit_path = argv[i + 1];        // circuits/image.r0c: the receipts carry an image proof
    else if (!strcmp(argv[i], "--image-code-object")) image_co_path = argv[i + 1];
    else if (!strcmp(argv[i], "--camt53-response")) camt["response"] = argv[i + 1];  // the library's own camt53 guest fed from an EBICS response
    else if (!strcmp(argv[i], "--pub-bank") || !strcmp(argv[i], "--pub-client") || !strcmp(argv[i], "--pub-witness") || !strcmp(argv[i], "--tx-key-raw") ||
             !strcmp(argv[i], "--witness-hex") || !strcmp(argv[i], "--iban") || !strcmp(argv[i], "--hostinfo") || !strcmp(argv[i], "--form"))
      camt[argv[i] + 2] = argv[i + 1];
    else { fprintf(stderr, "r0h_prove: unknown option %s\n", argv[i]); return 1; }
  }
  FILE* f = fopen(blob_path.c_str(), "rb");
  if (!f) { fprintf(stderr, "r0h_prove: cannot open %s\n", blob_path.c_str()); return 1; }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint32_t> blob((size_t)sz / 4);
  if (fread(blob.data(), 4, blob.size(), f) != blob.size()) { fprintf(stderr, "r0h_prove: short read\n"); return 1; }
  fclose(f);
  if (!elf_path.empty()) {  // prove(env, elf)
    auto slurp = [](const std::string& path, std::vector<uint8_t>* out) {
      FILE* g = fopen(path.c_str(), "rb");
      if (!g) return false;
      uint8_t buf[65536];
      for (size_t got; (got = fread(buf, 1, sizeof buf, g)) > 0;) out->insert(out->end(), buf, buf + got);
      fclose(g);
      return true;
    };
    std::vector<uint8_t> elf, raw;
    if (!slurp(elf_path, &elf)) { fprintf(stderr, "r0h_prove: cannot open %s\n", elf_path.c_str()); return 1; }
    if (!input_path.empty() && !slurp(input_path, &raw)) { fprintf(stderr, "r0h_prove: cannot open %s\n", input_path.c_str()); return 1; }
    if (raw.size() % 4) { fprintf(stderr, "r0h_prove: --input is a stream of 32-bit words\n"); return 1; }
    if (receipt_out.empty() && receipt_prefix.empty() && receipt_dir.empty()) { fprintf(stderr, "r0h_prove: --elf needs --receipt-out, --receipt-prefix or --receipt-dir\n"); return 1; }
    std::vector<uint32_t> words(raw.size() / 4);
    if (!words.empty()) memcpy(words.data(), raw.data(), raw.size());
    if (camt.count("response")) {
      // what the reference's `host` does between reading the pre-processed files and `prove` (host/src/main.rs:389-417), for this
      // library's camt53 guest: response.xml, the three public keys, the decrypted transaction key block, the witness signature, iban,
      // host info -> the guest's input words (r0h_camt53_guest_input)
      for (const char* need : {"pub-bank", "pub-client", "pub-witness", "tx-key-raw", "witness-hex", "iban"})
        if (!camt.count(need)) { fprintf(stderr, "r0h_prove: --camt53-response needs --%s\n", need); return 1; }
      std::map<std::string, std::vector<uint8_t>> file;
      for (const char* name : {"response", "pub-bank", "pub-client", "pub-witness", "tx-key-raw", "witness-hex"})
        if (!slurp(camt[name], &file[name])) { fprintf(stderr, "r0h_prove: cannot open %s\n", camt[name].c_str()); return 1; }
      r0h_ebics* eb = nullptr;
      CHECK(r0h_ebics_parse((const char*)file["response"].data(), file["response"].size(), &eb));
      uint32_t* stream = nullptr; size_t n_stream = 0;
      auto pem = [&](const char* k) { return (const char*)file[k].data(); };
      CHECK(r0h_camt53_guest_input(eb, pem("pub-bank"), file["pub-bank"].size(), pem("pub-client"), file["pub-client"].size(), pem("pub-witness"), file["pub-witness"].size(),
                                   file["tx-key-raw"].data(), file["tx-key-raw"].size(), pem("witness-hex"), file["witness-hex"].size(), camt["iban"].c_str(),
                                   camt.count("hostinfo") ? camt["hostinfo"].c_str() : "host:main", camt.count("form") ? (uint32_t)atoi(camt["form"].c_str()) : 1u, &stream, &n_stream));
      words.assign(stream, stream + n_stream);
      r0h_free_error((const char*)stream);
      r0h_ebics_free(eb);
    }
    // BASELINE.json configs[3] on the real workload: --receipts R sessions of this guest (host/src/main.rs:389-423 looped by
    // data/watchdog.sh:46-109), --contexts C of them in flight -- each on a context of its own with its own executor thread and
    // prover lanes -- one receipt file each, every receipt verified with the ELF afterwards (outside the timed region)
    if (contexts < 1 || contexts > 8) { fprintf(stderr, "r0h_prove: --contexts must be 1..8 with --elf\n"); return 1; }
    if (receipts > 1 && receipt_dir.empty()) { fprintf(stderr, "r0h_prove: --receipts R > 1 writes one file per receipt: use --receipt-dir\n"); return 1; }
    if (contexts > receipts) contexts = receipts;
    struct Worker { r0h_ctx* ctx = nullptr; r0h_circuit* circ = nullptr; r0h_circuit* image = nullptr; };
    std::vector<Worker> workers(contexts);
    std::vector<uint32_t> image_blob;
    if (!image_circuit_path.empty()) {
      std::vector<uint8_t> bytes;
      if (!slurp(image_circuit_path, &bytes) || bytes.size() % 4) { fprintf(stderr, "r0h_prove: cannot read %s\n", image_circuit_path.c_str()); return 1; }
      image_blob.resize(bytes.size() / 4);
      memcpy(image_blob.data(), bytes.data(), bytes.size());
    }
    for (Worker& w : workers) {
      CHECK(r0h_ctx_create((int)device, &w.ctx));
      CHECK(r0h_circuit_load(w.ctx, blob.data(), blob.size(), co_path.empty() ? nullptr : co_path.c_str(), &w.circ));
      if (!image_blob.empty()) {  // every receipt then carries an image proof: `receipt.verify(image_id)` needs no ELF
        CHECK(r0h_circuit_load(w.ctx, image_blob.data(), image_blob.size(), image_co_path.empty() ? nullptr : image_co_path.c_str(), &w.image));
        CHECK(r0h_ctx_set_image_circuit(w.ctx, w.image));
      }
    }
    r0h_ctx* ctx = workers[0].ctx;
    r0h_circuit* circ = workers[0].circ;
    std::vector<r0h_receipt*> made(receipts, nullptr);
    uint8_t image_id[32];
    uint64_t cycles = 0;
    std::atomic<unsigned> next_receipt{0};
    auto session_worker = [&](unsigned k) {
      for (unsigned u; (u = next_receipt.fetch_add(1)) < receipts;) {
        uint8_t id[32];
        uint64_t cyc = 0;
        CHECK(r0h_prove_elf(workers[k].ctx, workers[k].circ, elf.data(), elf.size(), words.data(), words.size(), po2, 0, &made[u], id, &cyc));
        if (u == 0) { memcpy(image_id, id, 32); cycles = cyc; }
      }
    };
    const auto t0 = std::chrono::steady_clock::now();
    {
      std::vector<std::thread> threads;
      for (unsigned k = 1; k < contexts; k++) threads.emplace_back(session_worker, k);
      session_worker(0);
      for (auto& t : threads) t.join();
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    r0h_receipt* rc = made[0];
    for (unsigned u = 0; u < receipts; u++) {
      char* text = nullptr;
      CHECK(r0h_receipt_to_json(made[u], &text));
      std::string path = receipt_out;
      if (receipts > 1 || (receipt_out.empty() && receipt_prefix.empty())) {
        char name[64];
        snprintf(name, sizeof name, "/receipt_%04u.json", u);
        path = receipt_dir + name;
      } else if (receipt_out.empty()) {  // where the reference's verifier looks: <camt53 file>-Receipt-<image id>-latest.json
        char id_hex[65];
        CHECK(r0h_image_id_to_hex(image_id, id_hex));
        path = receipt_prefix + "-Receipt-" + id_hex + "-latest.json";
      }
      if (u == 0) receipt_out = path;
      FILE* o = fopen(path.c_str(), "wb");
      if (!o || fwrite(text, 1, strlen(text), o) != strlen(text)) { fprintf(stderr, "r0h_prove: cannot write %s\n", path.c_str()); return 1; }
      fclose(o);
      r0h_free_error(text);
    }
    // the control root of every trace size the segments were proved at: what `r0h_verify --control-root` takes
    const size_t n_seg = r0h_receipt_n_segments(rc);
    std::vector<uint32_t> sizes;
    for (size_t i = 0; i < n_seg; i++) {
      const uint32_t* seal; size_t n_words;
      CHECK(r0h_receipt_segment(rc, i, &seal, &n_words, nullptr));
      int verdict = -1; uint32_t size = 0;
      CHECK(r0h_verify_seal(blob.data(), blob.size(), nullptr, nullptr, seal, n_words, &verdict, &size));
      if (verdict != R0H_VERIFY_OK) { fprintf(stderr, "r0h_prove: the verifier rejects seal %zu: %s\n", i, r0h_verify_reason(verdict)); return 3; }
      if (std::find(sizes.begin(), sizes.end(), size) == sizes.end()) sizes.push_back(size);
    }
    std::vector<uint32_t> root_table;
    for (size_t k = 0; k < sizes.size(); k++) {
      const size_t n = (size_t)1 << sizes[k];
      r0h_buf *code = nullptr, *data = nullptr;
      CHECK(r0h_buf_alloc(ctx, (size_t)r0h_circuit_group_size(circ, R0H_GROUP_CODE) * n * 4, &code));
      CHECK(r0h_buf_alloc(ctx, (size_t)r0h_circuit_group_size(circ, R0H_GROUP_DATA) * n * 4, &data));
      CHECK(r0h_witgen(ctx, circ, sizes[k], 0, code, data, nullptr));  // the circuit's fixed CODE columns
      uint32_t root[8];
      CHECK(r0h_code_root(ctx, code, r0h_circuit_group_size(circ, R0H_GROUP_CODE), sizes[k], root));
      root_table.push_back(sizes[k]);
      root_table.insert(root_table.end(), root, root + 8);
      CHECK(r0h_buf_free(code));
      CHECK(r0h_buf_free(data));
    }
    // every receipt verified the way the reference's verifier would (verifier/src/main.rs:124-126), with the ELF: host threads
    std::atomic<unsigned> next_check{0}, refused{0};
    const auto v0 = std::chrono::steady_clock::now();
    auto check_worker = [&] {
      for (unsigned u; (u = next_check.fetch_add(1)) < receipts;) {
        int verdict = -1;
        const char* e = r0h_receipt_verify_elf(made[u], blob.data(), blob.size(), root_table.data(), root_table.size() / 9, elf.data(), elf.size(), &verdict, nullptr, nullptr);
        if (e) { r0h_free_error(e); verdict = -1; }
        if (verdict == R0H_RECEIPT_V_OK && !image_blob.empty()) {  // ... and the way the reference's verifier does: the image id's 32 bytes alone
          e = r0h_receipt_verify_image(made[u], blob.data(), blob.size(), root_table.data(), root_table.size() / 9, image_blob.data(), image_blob.size(), nullptr, image_id, &verdict,
                                       nullptr, nullptr);
          if (e) { r0h_free_error(e); verdict = -1; }
        }
        if (verdict != R0H_RECEIPT_V_OK) { fprintf(stderr, "r0h_prove: receipt %u is refused: %s\n", u, r0h_receipt_verify_reason(verdict)); refused++; }
      }
    };
    {
      std::vector<std::thread> threads;
      for (unsigned k = 1; k < std::min(receipts, 8u); k++) threads.emplace_back(check_worker);
      check_worker();
      for (auto& t : threads) t.join();
    }
    const double verify_secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - v0).count();
    if (refused) return 3;
    char hex[65];
    CHECK(r0h_image_id_to_hex(image_id, hex));
    r0h_session_stats st;
    CHECK(r0h_last_session_stats(ctx, &st));
    printf("{\"receipt\": \"%s\", \"image_id\": \"%s\", \"receipts\": %u, \"contexts\": %u, \"segments\": %zu, \"cycles\": %llu, \"seconds\": %.4f, \"segments_per_s\": %.3f, \"receipts_per_s\": %.4f, "
           "\"executor_s\": %.4f, \"receipts_verified_with_the_elf\": %u, \"receipts_verified_with_the_image_id_alone\": %u, \"verify_seconds\": %.3f, \"control_roots\": [",
           receipt_out.c_str(), hex, receipts, contexts, n_seg, (unsigned long long)cycles, secs, (double)n_seg * receipts / secs, receipts / secs, st.executor_s, receipts,
           image_blob.empty() ? 0u : receipts, verify_secs);
    for (size_t k = 0; k < sizes.size(); k++) {
      const uint32_t* root = &root_table[9 * k + 1];
      printf("%s\"%u:%u,%u,%u,%u,%u,%u,%u,%u\"", k ? ", " : "", sizes[k], root[0], root[1], root[2], root[3], root[4], root[5], root[6], root[7]);
    }
    printf("]}\n");
    for (r0h_receipt* r : made) CHECK(r0h_receipt_free(r));
    for (Worker& w : workers) {
      CHECK(r0h_circuit_free(w.circ));
      if (w.image) { CHECK(r0h_ctx_set_image_circuit(w.ctx, nullptr)); CHECK(r0h_circuit_free(w.image)); }
      CHECK(r0h_ctx_destroy(w.ctx));
    }
    return 0;
  }
  if (contexts < 1 || contexts > 16) { fprintf(stderr, "r0h_prove: --contexts must be 1..16\n"); return 1; }
  if (receipts < 1 || (receipts > 1 && !receipt_out.empty())) { fprintf(stderr, "r0h_prove: --receipts R > 1 writes one file per receipt: use --receipt-dir\n"); return 1; }
  if (!receipt_out.empty() && !receipt_dir.empty()) { fprintf(stderr, "r0h_prove: --receipt-out and --receipt-dir exclude each other\n"); return 1; }
  const bool with_claims = !receipt_out.empty() || !receipt_dir.empty();
  // BASELINE.json configs[3]: a batch of independent receipts of `segments` segments each -- receipts x segments units on one
  // work queue, every lane takes the next unit when it is free
  const unsigned units = receipts * segments;

  // one context (+ circuit + resident witness) per in-flight lane, each driven by its own host thread: segments are
  // independent, so the lanes never talk to each other
  struct Lane {
    r0h_ctx* ctx = nullptr; r0h_circuit* circ = nullptr; r0h_buf* code = nullptr; r0h_buf* data = nullptr;
    std::vector<uint32_t> global, seal; size_t words = 0; unsigned proved = 0, loaded = 0;  // loaded: unit whose witness the buffers hold
    std::vector<std::pair<unsigned, std::vector<uint32_t>>> kept;  // (unit, seal) when receipts are to be written
  };
  std::vector<Lane> lanes(contexts);
  const size_t n = (size_t)1 << po2;
  for (unsigned k = 0; k < contexts; k++) {
    Lane& ln = lanes[k];
    CHECK(r0h_ctx_create((int)device, &ln.ctx));
    CHECK(r0h_circuit_load(ln.ctx, blob.data(), blob.size(), co_path.empty() ? nullptr : co_path.c_str(), &ln.circ));
    CHECK(r0h_buf_alloc(ln.ctx, (size_t)r0h_circuit_group_size(ln.circ, R0H_GROUP_CODE) * n * 4, &ln.code));
    CHECK(r0h_buf_alloc(ln.ctx, (size_t)r0h_circuit_group_size(ln.circ, R0H_GROUP_DATA) * n * 4, &ln.data));
    ln.global.resize(r0h_circuit_n_global(ln.circ) + 1);
    ln.seal.resize((size_t)1 << 20);
    CHECK(r0h_witgen(ln.ctx, ln.circ, po2, seed + k, ln.code, ln.data, ln.global.data()));
    ln.loaded = k;
  }
  // CODE is the same for every segment of this circuit and size: committed once, read by every lane's proofs
  r0h_code_commit* code_commit = nullptr;
  CHECK(r0h_code_commit_new(lanes[0].ctx, lanes[0].code, r0h_circuit_group_size(lanes[0].circ, R0H_GROUP_CODE), po2, &code_commit));
  // the claims of every unit (receipt r, segment s): unit = r * segments + s
  std::vector<uint8_t> journal(journal_text.size() + 8);
  size_t jn = 0;
  CHECK(r0h_serde_encode_str((const uint8_t*)journal_text.data(), journal_text.size(), journal.data(), journal.size(), &jn));
  journal.resize(jn);
  std::vector<r0h_receipt_claim> claims(units);
  std::vector<std::string> image_ids(receipts);
  if (with_claims) {
    if (r0h_circuit_n_global(lanes[0].circ) < 8) { fprintf(stderr, "r0h_prove: the circuit exposes fewer than 8 public inputs: a claim cannot be bound to its seals\n"); return 1; }
    for (unsigned r = 0; r < receipts; r++) {
      std::vector<r0h_system_state> st(segments + 1);
      for (unsigned k = 0; k <= segments; k++) {
        char name[96];
        int nn = snprintf(name, sizeof name, "r0hip synthetic session %llu/%u/%u", seed, r, k);
        st[k].pc = 0;
        CHECK(r0h_sha256((const uint8_t*)name, (size_t)nn, st[k].merkle_root));
      }
      for (unsigned k = 0; k < segments; k++) {
        r0h_receipt_claim& c = claims[r * segments + k];
        memset(&c, 0, sizeof c);
        c.pre = st[k];
        c.post = st[k + 1];
        const bool last = k + 1 == segments;
        c.exit_system = last ? 0 : 2;  // Halted(0) | SystemSplit
        if (last) CHECK(r0h_output_digest(journal.data(), journal.size(), nullptr, c.output_digest));
      }
      uint8_t id[32];
      CHECK(r0h_system_state_digest(&st[0], id));
      char hex[65];
      CHECK(r0h_image_id_to_hex(id, hex));  // the reference's IMAGE_ID.hex convention: eight {:08x} words
      image_ids[r] = hex;
    }
  }
  std::atomic<unsigned> next_unit{0};
  auto work = [&](unsigned k) {
    Lane& ln = lanes[k];
    for (unsigned u; (u = next_unit.fetch_add(1)) < units;) {
      const unsigned s = u % segments;  // segment index within its receipt
      if (with_claims) {  // the claim's naming words are the segment's public inputs
        uint8_t cd[32];
        CHECK(r0h_claim_digest(&claims[u], cd));
        std::fill(ln.global.begin(), ln.global.end(), 0u);
        CHECK(r0h_claim_globals(cd, ln.global.data()));
        CHECK(r0h_witgen_public(ln.ctx, ln.circ, po2, seed + u, ln.global.data(), ln.code, ln.data));
        ln.loaded = ~0u;
      } else if (u != ln.loaded) {
        CHECK(r0h_witgen(ln.ctx, ln.circ, po2, seed + u, ln.code, ln.data, ln.global.data()));
        ln.loaded = u;
      }
      CHECK(r0h_prove_segment_committed(ln.ctx, ln.circ, po2, code_commit, ln.data, ln.global.data(), ln.seal.data(), ln.seal.size(), &ln.words));
      ln.proved++;
      (void)s;
      if (with_claims) ln.kept.emplace_back(u, std::vector<uint32_t>(ln.seal.begin(), ln.seal.begin() + ln.words));
    }
  };
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> threads;
  for (unsigned k = 1; k < contexts; k++) threads.emplace_back(work, k);
  work(0);
  for (auto& t : threads) t.join();
  const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  // report the seal lane 0 proved last (with one lane: the last unit; with several, whichever unit the queue handed it)
  const Lane& l0 = lanes[0];
  uint64_t h = 1469598103934665603ull;  // FNV-1a, to compare runs
  for (size_t i = 0; i < l0.words; i++) { h ^= l0.seal[i]; h *= 1099511628211ull; }
  printf("{\"segments\": %u, \"receipts\": %u, \"contexts\": %u, \"po2\": %u, \"seal_words\": %zu, \"seal_fnv1a\": \"%016llx\", \"seconds\": %.6f, "
         "\"segments_per_s\": %.4f, \"receipts_per_s\": %.4f}\n",
         segments, receipts, contexts, po2, l0.words, (unsigned long long)h, total, units / total, receipts / total);
  if (verify) {  // outside the timed region: the host-side verifier on every lane's last seal
    for (const Lane& ln : lanes) {
      if (!ln.proved) continue;
      int verdict = -1;
      CHECK(r0h_verify_seal(blob.data(), blob.size(), nullptr, nullptr, ln.seal.data(), ln.words, &verdict, nullptr));
      if (verdict != R0H_VERIFY_OK) { fprintf(stderr, "r0h_prove: the verifier rejects the seal: %s\n", r0h_verify_reason(verdict)); return 3; }
    }
    fprintf(stderr, "r0h_prove: seals verified\n");
  }
  if (with_claims) {
    // the Receipt JSON `host` writes (host/src/main.rs:251-252, 299-316): all segment seals in order + the journal, which for the
    // hyperfridge guest is the serde word stream of the committed JSON string (host/src/main.rs:258-267)
    uint32_t root[8];
    CHECK(r0h_code_commit_root(code_commit, root));
    printf("{\"control_root\": {\"po2\": %u, \"root\": [%u, %u, %u, %u, %u, %u, %u, %u]}, \"image_ids\": [", po2, root[0], root[1], root[2], root[3], root[4], root[5], root[6], root[7]);
    for (unsigned r = 0; r < receipts; r++) printf("%s\"%s\"", r ? ", " : "", image_ids[r].c_str());
    printf("]}\n");
    for (unsigned r = 0; r < receipts; r++) {
      r0h_receipt* rc = nullptr;
      CHECK(r0h_receipt_new(R0H_RECEIPT_COMPOSITE, journal.data(), journal.size(), &rc));
      for (unsigned k = 0; k < segments; k++)
        for (const Lane& ln : lanes)
          for (const auto& kv : ln.kept)
            if (kv.first == r * segments + k) CHECK(r0h_receipt_add_segment_claim(rc, kv.second.data(), kv.second.size(), k, &claims[kv.first], nullptr));
      if (r0h_receipt_n_segments(rc) != segments) { fprintf(stderr, "r0h_prove: receipt %u is missing segments\n", r); return 3; }
      char* text = nullptr;
      CHECK(r0h_receipt_to_json(rc, &text));
      char name[64];
      snprintf(name, sizeof name, "/receipt_%04u.json", r);
      const std::string path = receipt_out.empty() ? receipt_dir + name : receipt_out;
      FILE* o = fopen(path.c_str(), "wb");
      if (!o || fwrite(text, 1, strlen(text), o) != strlen(text)) { fprintf(stderr, "r0h_prove: cannot write %s\n", path.c_str()); return 1; }
      fclose(o);
      r0h_free_error(text);
      CHECK(r0h_receipt_free(rc));
    }
  }
  if (!seal_out.empty()) {
    FILE* o = fopen(seal_out.c_str(), "wb");
    if (!o || fwrite(l0.seal.data(), 4, l0.words, o) != l0.words) { fprintf(stderr, "r0h_prove: cannot write %s\n", seal_out.c_str()); return 1; }
    fclose(o);
  }
  CHECK(r0h_code_commit_free(code_commit));
  for (Lane& ln : lanes) {
    CHECK(r0h_buf_free(ln.code));
    CHECK(r0h_buf_free(ln.data));
    CHECK(r0h_circuit_free(ln.circ));
    CHECK(r0h_ctx_destroy(ln.ctx));
  }
  return 0;
}
