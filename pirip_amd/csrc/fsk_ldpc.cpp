// pirip_amd/csrc/fsk_ldpc.cpp -- see fsk_ldpc.hpp. Host-only (the Tx-side framer is a CPU tool in the reference too).
#include "fsk_ldpc.hpp"

#include <algorithm>
#include <cstdio>
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
