// pirip_amd/csrc/fsk_ldpc.cpp -- see fsk_ldpc.hpp. Host-only (the Tx-side framer is a CPU tool in the reference too).
#include "fsk_ldpc.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace pirip {

std::string LdpcCode::load(const std::string &path)
{
    std::ifstream f(path);
    if (!f) return "cannot open " + path;
    std::string line, key;
    int rows = -1;
    bool have_uw = false;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls(line);
        ls >> key;
        if (key == "name") ls >> name;
        else if (key == "n") ls >> n;
        else if (key == "k") ls >> k;
        else if (key == "max_iter") ls >> max_iter;
        else if (key == "uw_thresh1") ls >> uw_thresh1;
        else if (key == "uw_thresh2") ls >> uw_thresh2;
        else if (key == "bad_uw_thresh") ls >> bad_uw_thresh;
        else if (key == "llr_map") {
            std::string v; ls >> v;
            if (v == "upstream") llr_map = 0; else if (v == "rician") llr_map = 1; else return "llr_map must be upstream or rician, not '" + v + "'";
        }
        else if (key == "uw") {
            for (int i = 0; i < kUwBits; i++) { int b = -1; ls >> b; if (b != 0 && b != 1) return "uw needs 32 bits"; uw[i] = (uint8_t)b; }
            have_uw = true;
        } else if (key == "rows") { ls >> rows; break; }
        else return "unknown key '" + key + "'";
    }
    m = n - k;
    if (n <= 0 || k <= 0 || m <= 0 || (k % 8) || k < 24 || rows != m || !have_uw || max_iter < 1) return "bad header (n, k, rows, uw, max_iter)";
    if (n > 4096) return "codeword too long for the decoder kernel (n <= 4096)";
    if (m < k / 8) return "fewer parity bits than payload bytes (the decoder packs the payload into the parity area: n - k >= k/8)";
    row_ptr.assign(1, 0);
    col_idx.clear();
    for (int r = 0; r < m; r++) {
        if (!std::getline(f, line)) return "too few rows";
        std::istringstream ls(line);
        std::vector<int32_t> cols;
        int c;
        while (ls >> c) { if (c < 0 || c >= n) return "column index out of range"; cols.push_back(c); }
        std::sort(cols.begin(), cols.end());
        if (cols.empty() || std::adjacent_find(cols.begin(), cols.end()) != cols.end()) return "empty row or duplicate column";
        if (cols.size() > 64) return "row weight above 64";
        col_idx.insert(col_idx.end(), cols.begin(), cols.end());
        row_ptr.push_back((int32_t)col_idx.size());
    }
    // CSC view: edges of each column in ascending row order (edge index = position in col_idx)
    std::vector<int32_t> cnt(n + 1, 0);
    for (int32_t c : col_idx) cnt[c + 1]++;
    col_ptr.assign(n + 1, 0);
    for (int i = 0; i < n; i++) col_ptr[i + 1] = col_ptr[i] + cnt[i + 1];
    col_edge.assign(col_idx.size(), 0);
    std::vector<int32_t> fill(col_ptr.begin(), col_ptr.end() - 1);
    for (int r = 0; r < m; r++)
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; e++) col_edge[fill[col_idx[e]]++] = e;
    for (int i = 0; i < n; i++) if (col_ptr[i + 1] == col_ptr[i]) return "column without a parity check";
    // accumulator shape: row p holds parity column k+p, and k+p-1 for p > 0, and no other parity column
    accumulator = true;
    for (int r = 0; r < m && accumulator; r++) {
        int npar = 0;
        bool own = false, prev = false;
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
            const int c = col_idx[e];
            if (c >= k) { npar++; own |= (c == k + r); prev |= (r > 0 && c == k + r - 1); }
        }
        accumulator = own && (r == 0 ? npar == 1 : (prev && npar == 2));
    }
    return "";
}

void LdpcCode::encode(const uint8_t *data_bits, uint8_t *parity_bits) const
{
    int prev = 0;
    for (int p = 0; p < m; p++) {
        int par = 0;
        for (int e = row_ptr[p]; e < row_ptr[p + 1]; e++)
            if (col_idx[e] < k) par += data_bits[col_idx[e]] & 1;
        prev = (par + prev) & 1;
        parity_bits[p] = (uint8_t)prev;
    }
}

int LdpcCode::parity_checks_ok(const uint8_t *cw) const
{
    int ok = 0;
    for (int r = 0; r < m; r++) {
        int x = 0;
        for (int e = row_ptr[r]; e < row_ptr[r + 1]; e++) x ^= cw[col_idx[e]] & 1;
        ok += !x;
    }
    return ok;
}

uint16_t crc16_ccitt(const uint8_t *bytes, int n)
{
    uint16_t crc = 0xFFFF;
    while (n--) {
        uint8_t x = (uint8_t)(crc >> 8) ^ *bytes++;
        x ^= x >> 4;
        crc = (uint16_t)((crc << 8) ^ ((uint16_t)x << 12) ^ ((uint16_t)x << 5) ^ (uint16_t)x);
    }
    return crc;
}

void pack_bits_msb(uint8_t *bytes, const uint8_t *bits, int nbits)
{
    std::memset(bytes, 0, (size_t)(nbits + 7) / 8);
    for (int i = 0; i < nbits; i++) bytes[i >> 3] |= (uint8_t)((bits[i] & 1) << (7 - (i & 7)));
}

void unpack_bits_msb(uint8_t *bits, const uint8_t *bytes, int nbits)
{
    for (int i = 0; i < nbits; i++) bits[i] = (bytes[i >> 3] >> (7 - (i & 7))) & 1;
}

void insert_crc(uint8_t *data_bits, int k)
{
    std::vector<uint8_t> bytes((size_t)k / 8);
    pack_bits_msb(bytes.data(), data_bits, k - 16);
    const uint16_t crc = crc16_ccitt(bytes.data(), k / 8 - 2);
    const uint8_t cb[2] = {(uint8_t)(crc >> 8), (uint8_t)(crc & 0xff)};
    unpack_bits_msb(data_bits + k - 16, cb, 16);
}

void frame_bits(const LdpcCode &c, const uint8_t *data_bits, uint8_t *frame)
{
    std::memcpy(frame, c.uw, kUwBits);
    std::memcpy(frame + kUwBits, data_bits, (size_t)c.k);
    c.encode(data_bits, frame + kUwBits + c.k);
}

std::vector<uint8_t> preamble_bits(int M)
{
    const int nsym = 50 * (M >> 1), nbits = nsym * (M >> 1);
    std::vector<uint8_t> b((size_t)nbits);
    int sym = 0;
    for (int i = 0; i + 1 < nbits; i += 2) { b[i] = (sym >> 1) & 1; b[i + 1] = sym & 1; sym++; }
    if (nbits & 1) b[nbits - 1] = 0;
    return b;
}

void testframe_payload(uint8_t *data_bits, int k)
{
    uint32_t s = 1;
    for (int i = 0; i < k; i++) { s = (s * 1103515245u + 12345u) & 0x7fffffffu; data_bits[i] = (uint8_t)((s >> 16) & 1u); }
}

}  // namespace pirip

namespace pirip {

namespace {
// Bank-conflict bookkeeping of the decoder's two gathers for a layout (pi: variable -> storage index, rho: row -> position).
//   check pass:    lanes of a 32-group are the rows at positions 32 g .. 32 g + 31; gather j reads Q[pi[col_j]]      -> bank pi mod 32
//   variable pass: lanes are the variables at storage indices 32 g .. 32 g + 31; gather t reads message (slot, rho[row]) -> bank rho mod 32
// cell counts per (pass, group, slot, bank); cost = sum over cells of (count - 1)+ = extra LDS cycles per decoder iteration.
struct LayoutSearch {
    const LdpcCode &c;
    std::vector<int> pi, rho, cnt, erow, eslot, etpos;
    int cost = 0;
    size_t off2;
    explicit LayoutSearch(const LdpcCode &code) : c(code)
    {
        pi.resize(kFastVars); rho.resize(kFastRows);
        for (int i = 0; i < kFastVars; i++) pi[(size_t)i] = i;
        for (int i = 0; i < kFastRows; i++) rho[(size_t)i] = i;
        const size_t E = c.col_idx.size();
        erow.resize(E); eslot.resize(E); etpos.resize(E);
        for (int r = 0; r < c.m; r++)
            for (int e = c.row_ptr[r], j = 0; e < c.row_ptr[r + 1]; e++, j++) { erow[(size_t)e] = r; eslot[(size_t)e] = j; }
        for (int v = 0; v < c.n; v++)
            for (int x = c.col_ptr[v], t = 0; x < c.col_ptr[v + 1]; x++, t++) etpos[(size_t)c.col_edge[x]] = t;
        off2 = (size_t)(kFastRows / 32) * kFastRowDeg * 32;
        cnt.assign(off2 + (size_t)(kFastVars / 32) * kFastColDeg * 32, 0);
        for (size_t e = 0; e < E; e++) edge(e, +1);
    }
    void bump(size_t cell, int d)
    {
        int &x = cnt[cell];
        if (d > 0) { if (x >= 1) cost++; x++; } else { if (x >= 2) cost--; x--; }
    }
    void edge(size_t e, int d)                      // the two cells edge e occupies under the current pi / rho
    {
        const int r = erow[e], v = c.col_idx[e];
        bump((size_t)(((rho[(size_t)r] >> 5) * kFastRowDeg + eslot[e]) * 32 + (pi[(size_t)v] & 31)), d);
        bump(off2 + (size_t)(((pi[(size_t)v] >> 5) * kFastColDeg + etpos[e]) * 32 + (rho[(size_t)r] & 31)), d);
    }
    void var_edges(int v, int d) { if (v < c.n) for (int x = c.col_ptr[v]; x < c.col_ptr[v + 1]; x++) edge((size_t)c.col_edge[x], d); }
    void row_edges(int r, int d) { if (r < c.m) for (int e = c.row_ptr[r]; e < c.row_ptr[r + 1]; e++) edge((size_t)e, d); }
    // swap the storage of two variables (or the positions of two rows); returns the new cost
    int swap_vars(int a, int b) { var_edges(a, -1); var_edges(b, -1); std::swap(pi[(size_t)a], pi[(size_t)b]); var_edges(a, +1); var_edges(b, +1); return cost; }
    int swap_rows(int a, int b)
    {
        // a variable that sits in both rows would be touched twice: take the rows' edges out one row at a time
        row_edges(a, -1); row_edges(b, -1); std::swap(rho[(size_t)a], rho[(size_t)b]); row_edges(a, +1); row_edges(b, +1);
        return cost;
    }
};
}  // namespace

DecoderLayout make_decoder_layout(const LdpcCode &c)
{
    DecoderLayout L;
    int maxdeg = 0, maxcol = 0;
    for (int r = 0; r < c.m; r++) maxdeg = std::max(maxdeg, (int)(c.row_ptr[r + 1] - c.row_ptr[r]));
    for (int v = 0; v < c.n; v++) maxcol = std::max(maxcol, (int)(c.col_ptr[v + 1] - c.col_ptr[v]));
    L.maxdeg = maxdeg; L.maxcol = maxcol;
    if (c.m > kFastRows || c.n > kFastVars || maxdeg > kFastRowDeg || maxcol > kFastColDeg) return L;
    LayoutSearch S(c);
    L.gather_conflicts_identity = S.cost;
    // Annealed descent over swaps of two variables' storage indices / two rows' positions. Deterministic generator: the layout
    // is a function of the code alone. A worsening swap is kept with a probability that falls to zero over the run.
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](int n) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (int)((s >> 11) % (uint64_t)n); };
    const int tries = getenv("PIRIP_LAYOUT_TRIES") ? atoi(getenv("PIRIP_LAYOUT_TRIES")) : 600000;
    int best = S.cost;
    std::vector<int> best_pi = S.pi, best_rho = S.rho;
    // who sits at each storage index / position (-1: nobody)
    std::vector<int> var_at(kFastVars, -1), row_at(kFastRows, -1);
    for (int v = 0; v < kFastVars; v++) var_at[(size_t)S.pi[(size_t)v]] = v;
    for (int r = 0; r < kFastRows; r++) row_at[(size_t)S.rho[(size_t)r]] = r;
    const int E = (int)c.col_idx.size();
    for (int it = 0; it < tries && best > 0; it++) {
        // A move starts from a random edge. If one of its two cells is crowded on the edge's bank, the edge's variable (or row) trades
        // places with one that sits on a bank the cell does not use yet; otherwise (one try in eight) two random variables / rows swap.
        const int e = rnd(E);
        const int r = S.erow[(size_t)e], v = c.col_idx[(size_t)e];
        const size_t cell1 = (size_t)(((S.rho[(size_t)r] >> 5) * kFastRowDeg + S.eslot[(size_t)e]) * 32);
        const size_t cell2 = S.off2 + (size_t)(((S.pi[(size_t)v] >> 5) * kFastColDeg + S.etpos[(size_t)e]) * 32);
        const bool crowded1 = S.cnt[cell1 + (size_t)(S.pi[(size_t)v] & 31)] > 1, crowded2 = S.cnt[cell2 + (size_t)(S.rho[(size_t)r] & 31)] > 1;
        bool vars;
        int a, b;
        if (crowded1 || crowded2) {
            vars = crowded1 && (!crowded2 || rnd(2));
            const size_t cell = vars ? cell1 : cell2;
            int freeb = -1;
            for (int k = 0, b0 = rnd(32); k < 32; k++) if (S.cnt[cell + (size_t)((b0 + k) & 31)] == 0) { freeb = (b0 + k) & 31; break; }
            if (freeb < 0) continue;
            a = vars ? v : r;
            b = vars ? var_at[(size_t)(freeb + 32 * rnd(kFastVars / 32))] : row_at[(size_t)(freeb + 32 * rnd(kFastRows / 32))];
        } else {
            if (rnd(8)) continue;
            vars = rnd(3) != 0;
            const int n = vars ? kFastVars : kFastRows;
            a = rnd(n); b = rnd(n);
        }
        if (a == b || a < 0 || b < 0) continue;
        const int before = S.cost;
        const int after = vars ? S.swap_vars(a, b) : S.swap_rows(a, b);
        const int worse = after - before;
        // temperature: accept +1 with probability ~ 1/8 at the start, never in the last third
        const bool accept = worse <= 0 || (it < tries * 2 / 3 && worse == 1 && rnd(8 + 40 * it / (tries / 3 + 1)) == 0);
        if (!accept) { if (vars) S.swap_vars(a, b); else S.swap_rows(a, b); continue; }
        if (vars) { var_at[(size_t)S.pi[(size_t)a]] = a; var_at[(size_t)S.pi[(size_t)b]] = b; } else { row_at[(size_t)S.rho[(size_t)a]] = a; row_at[(size_t)S.rho[(size_t)b]] = b; }
        if (S.cost < best) { best = S.cost; best_pi = S.pi; best_rho = S.rho; }
    }
    const std::vector<int> &pi = best_pi, &rho = best_rho;
    L.gather_conflicts = best;
    L.rcol.assign((size_t)kFastRows * kFastRowDeg, 0xFFFFu);
    L.vedge.assign((size_t)kFastVars * kFastColDeg, 0xFFFFu);
    L.vsrc.assign((size_t)kFastVars, 0xFFFFu);
    for (int v = 0; v < c.n; v++) L.vsrc[(size_t)pi[(size_t)v]] = (uint16_t)v;
    for (int r = 0; r < c.m; r++)
        for (int e = c.row_ptr[r], j = 0; e < c.row_ptr[r + 1]; e++, j++)
            L.rcol[(size_t)rho[(size_t)r] * kFastRowDeg + j] = (uint16_t)pi[(size_t)c.col_idx[e]];
    for (int v = 0; v < c.n; v++)
        for (int x = c.col_ptr[v], t = 0; x < c.col_ptr[v + 1]; x++, t++) {
            const int e = c.col_edge[x];
            L.vedge[(size_t)pi[(size_t)v] * kFastColDeg + t] = (uint16_t)(S.eslot[(size_t)e] * kFastRows + rho[(size_t)S.erow[(size_t)e]]);
        }
    L.ok = true;
    return L;
}

}  // namespace pirip
