// TEST INFRASTRUCTURE - NOT PRODUCT CODE. See peptides_oracle.hpp.
#include "peptides_oracle.hpp"

#include <map>
#include <sstream>

#include "../microphaser_amd/csrc/model.hpp"

namespace mp_oracle {

namespace {
const std::map<std::string, char>& pairs() {  // make_pairs (src/peptides.rs:85-117)
    static const std::map<std::string, char> m = [] {
        const std::pair<char, std::vector<const char*>> grouped[] = {
            {'I', {"ATT", "ATC", "ATA"}}, {'L', {"CTT", "CTC", "CTA", "CTG", "TTA", "TTG"}}, {'V', {"GTT", "GTC", "GTA", "GTG"}},
            {'F', {"TTT", "TTC"}}, {'M', {"ATG"}}, {'C', {"TGT", "TGC"}}, {'A', {"GCT", "GCC", "GCA", "GCG"}},
            {'G', {"GGT", "GGC", "GGA", "GGG"}}, {'P', {"CCT", "CCC", "CCA", "CCG"}}, {'T', {"ACT", "ACC", "ACA", "ACG"}},
            {'S', {"TCT", "TCC", "TCA", "TCG", "AGT", "AGC"}}, {'Y', {"TAT", "TAC"}}, {'W', {"TGG"}}, {'Q', {"CAA", "CAG"}},
            {'N', {"AAT", "AAC"}}, {'H', {"CAT", "CAC"}}, {'E', {"GAA", "GAG"}}, {'D', {"GAT", "GAC"}}, {'K', {"AAA", "AAG"}},
            {'R', {"CGT", "CGC", "CGA", "CGG", "AGA", "AGG"}}, {'X', {"TAA", "TAG", "TGA"}}};
        std::map<std::string, char> out;
        for (const auto& g : grouped)
            for (const char* c : g.second) out[c] = g.first;
        return out;
    }();
    return m;
}
char complement(char c) {  // bio::alphabets::dna::complement on upper-case input
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'N': return 'N'; case 'R': return 'Y'; case 'Y': return 'R'; case 'S': return 'S'; case 'W': return 'W';
        case 'K': return 'M'; case 'M': return 'K'; case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
        default: return c;
    }
}
}  // namespace

std::string to_protein(const std::string& s, int frame) {
    std::string r;
    for (char c : s) r.push_back((c >= 'a' && c <= 'z') ? char(c - 32) : c);
    if (frame < 0) {
        std::string rc(r.rbegin(), r.rend());
        for (char& c : rc) c = complement(c);
        r = rc;
        frame = -frame;
    }
    std::string p;
    if (r.size() < 2) throw mp::Error("reference would panic: attempt to subtract with overflow (to_protein)");
    for (size_t i = size_t(frame) - 1; i < r.size() - 2; i += 3) {
        auto it = pairs().find(r.substr(i, 3));
        if (it == pairs().end()) throw mp::Error("reference would panic: called `Result::unwrap()` on an `Err` value (codon " + r.substr(i, 3) + ")");
        p.push_back(it->second);
    }
    return p;
}

std::string build_reference(const std::string& fasta_text, size_t peptide_length, std::set<std::string>& set) {
    std::string out;
    std::istringstream in(fasta_text);
    std::string line, id, seq;
    bool have = false;
    auto flush = [&]() {
        if (!have) return;
        int frame = (!id.empty() && id.back() == 'F') ? 1 : -1;  // :161-164
        size_t base_length = peptide_length * 3;
        for (size_t i = 0; i + base_length <= seq.size(); i += 3) {  // :168-174
            std::string pep = to_protein(seq.substr(i, base_length), frame);
            out += ">" + id + "\n" + pep + "\n";
            set.insert(pep);
        }
    };
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == '>') {
            flush();
            size_t sp = line.find_first_of(" \t");
            id = line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
            seq.clear();
            have = true;
        } else {
            seq += line;
        }
    }
    flush();
    return out;
}

std::string bincode_set(const std::set<std::string>& set) {
    std::string b;
    auto u64 = [&](uint64_t v) { for (int i = 0; i < 8; i++) b.push_back(char(v >> (8 * i))); };
    u64(set.size());
    for (const auto& s : set) { u64(s.size()); b += s; }
    return b;
}

}  // namespace mp_oracle
