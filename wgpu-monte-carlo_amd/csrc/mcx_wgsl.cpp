// mcx_wgsl.cpp -- scalar WGSL function strings -> HIP C++ device functions (include/mcx.h: mcx_wgsl_translate).
//
// The reference's native half takes its integrands as WGSL text -- the output of its Python transpiler, user-written
// strings, and the importance-sampling wrappers its Python half generates (python/wgpu_montecarlo/__init__.py:740-742,
// 893-905, 968-980) -- and splices them into a WGSL shader compiled by naga (src/shader_gen.rs:45-128). libmcx's kernels
// are HIP, so the same text is translated here: a tokenizer and a recursive-descent parser over the scalar subset those
// sources use (f32 / i32 / u32 / bool, let / var / const, if / else, for / while / loop, helper functions, the WGSL
// builtins), emitting one `MCX_DEV` function per WGSL function. No vectors, matrices, structs, pointers or textures:
// those are refused by name.
//
// math (as wgpu_montecarlo/emit_hip.py): 0 "precise" = the ocml routines; 1 "default" = the hardware exp / log / sqrt, the
// range-reduced hardware sin / cos / tan, pow as exp2(y log2|x|), sinh / cosh on the hardware exp (device/mcx_device.hpp);
// 2 "fast" = sin / cos / tan as the bare instructions. `/` stays the C operator in every mode: the translator does not type
// expressions, and an integer quotient must stay one. A literal whole exponent -- `pow(x, 2.0)` is what the reference's
// transpiler writes for x**2 -- becomes the product chain McxPowI<n> (prelude: mcx_wgsl_prelude()) in every mode.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mcx.h"
#include "mcx_internal.hpp"

namespace {

struct TranslateError { std::string msg; };
[[noreturn]] void bad(const std::string& m) { throw TranslateError{m}; }

enum Kind { NUM, ID, OP, END };
struct Token { Kind kind; std::string text; size_t pos = 0; };

const char* const kOps3[] = {"<<=", ">>="};
const char* const kOps2[] = {"->", "<<", ">>", "<=", ">=", "==", "!=", "&&", "||", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "++", "--"};
const char kOps1[] = "-+*/%<>=!&|^~(){}[],;:.@";

std::vector<Token> tokenize(const std::string& src) {
    std::string text = src;
    while (!text.empty() && isspace((unsigned char)text.back())) text.pop_back();
    std::vector<Token> out;
    size_t i = 0;
    const size_t n = text.size();
    auto digit = [&](size_t j) { return j < n && isdigit((unsigned char)text[j]); };
    while (i < n) {
        while (i < n && isspace((unsigned char)text[i])) ++i;
        if (i >= n) break;
        if (text.compare(i, 2, "//") == 0) {                                  // line comment
            while (i < n && text[i] != '\n') ++i;
            continue;
        }
        if (text.compare(i, 2, "/*") == 0) {                                  // block comment (must be closed, else it is an operator run)
            const size_t end = text.find("*/", i + 2);
            if (end != std::string::npos) { i = end + 2; continue; }
        }
        const size_t start = i;
        if (text[i] == '0' && i + 1 < n && (text[i + 1] == 'x' || text[i + 1] == 'X') && i + 2 < n && isxdigit((unsigned char)text[i + 2])) {
            i += 2;
            while (i < n && isxdigit((unsigned char)text[i])) ++i;
            if (i < n && (text[i] == 'i' || text[i] == 'u')) ++i;
            out.push_back({NUM, text.substr(start, i - start), start});
            continue;
        }
        if (digit(i) || (text[i] == '.' && digit(i + 1))) {
            while (digit(i)) ++i;
            if (i < n && text[i] == '.') { ++i; while (digit(i)) ++i; }
            if (i < n && (text[i] == 'e' || text[i] == 'E')) {                 // exponent only if digits follow
                size_t j = i + 1;
                if (j < n && (text[j] == '+' || text[j] == '-')) ++j;
                if (digit(j)) { i = j; while (digit(i)) ++i; }
            }
            if (i < n && (text[i] == 'f' || text[i] == 'h' || text[i] == 'i' || text[i] == 'u')) ++i;
            out.push_back({NUM, text.substr(start, i - start), start});
            continue;
        }
        if (isalpha((unsigned char)text[i]) || text[i] == '_') {
            while (i < n && (isalnum((unsigned char)text[i]) || text[i] == '_')) ++i;
            out.push_back({ID, text.substr(start, i - start), start});
            continue;
        }
        bool matched = false;
        for (const char* op : kOps3) if (text.compare(i, 3, op) == 0) { out.push_back({OP, op, i}); i += 3; matched = true; break; }
        if (matched) continue;
        for (const char* op : kOps2) if (text.compare(i, 2, op) == 0) { out.push_back({OP, op, i}); i += 2; matched = true; break; }
        if (matched) continue;
        if (strchr(kOps1, text[i]) && text[i] != '\0') { out.push_back({OP, std::string(1, text[i]), i}); ++i; continue; }
        bad("WGSL function string: cannot tokenize near '" + text.substr(i, 20) + "'");
    }
    return out;
}

std::string number(const std::string& text) {
    if (text.size() > 1 && text[0] == '0' && (text[1] == 'x' || text[1] == 'X')) {
        std::string body = text;
        const bool u = body.back() == 'u';
        while (!body.empty() && (body.back() == 'i' || body.back() == 'u')) body.pop_back();
        return body + (u ? "u" : "");
    }
    char suffix = 0;
    const char last = text.back();
    if (last == 'f' || last == 'h' || last == 'i' || last == 'u') suffix = last;
    std::string body = suffix ? text.substr(0, text.size() - 1) : text;
    bool is_float = suffix == 'f' || suffix == 'h';
    for (char c : body) if (c == '.' || c == 'e' || c == 'E') is_float = true;
    if (is_float) {
        bool has_point_or_exp = false;
        for (char c : body) if (c == '.' || c == 'e' || c == 'E') has_point_or_exp = true;
        if (!has_point_or_exp) body += ".0";
        return body + "f";
    }
    return body + (suffix == 'u' ? "u" : "");
}

// `2.0f`, `(-3.0f)`, `4`: a whole number of magnitude <= 64, else false
bool whole_exponent(const std::string& text_in, int* value) {
    std::string t;
    for (char c : text_in) if (c != ' ') t += c;
    int open = 0, close = 0;
    for (char c : t) { open += c == '('; close += c == ')'; }
    if (open != close) return false;
    size_t i = 0;
    while (i < t.size() && t[i] == '(') ++i;
    bool neg = false;
    if (i < t.size() && t[i] == '-') { neg = true; ++i; }
    while (i < t.size() && t[i] == '(') ++i;
    const size_t d0 = i;
    while (i < t.size() && isdigit((unsigned char)t[i])) ++i;
    if (i == d0) return false;
    const std::string digits = t.substr(d0, i - d0);
    if (i < t.size() && t[i] == '.') { ++i; while (i < t.size() && t[i] == '0') ++i; }
    if (i < t.size() && t[i] == 'f') ++i;
    while (i < t.size() && t[i] == ')') ++i;
    if (i != t.size() || digits.size() > 4) return false;
    const int n = atoi(digits.c_str());
    if (n > 64) return false;
    *value = neg ? -n : n;
    return true;
}

const std::map<std::string, std::string> kTypes = {{"f32", "float"}, {"f16", "float"}, {"i32", "int"}, {"u32", "unsigned int"}, {"bool", "bool"}};
const std::map<std::string, std::string> kTableCalls = {{"pdf_target_from_table", "mcx_user_pdf_target"},
                                                        {"pdf_proposal_from_table", "mcx_user_pdf_proposal"}};
std::map<std::string, std::string> builtins_for(int math) {
    std::map<std::string, std::string> b = {
        {"abs", "fabsf"}, {"sin", "sinf"}, {"cos", "cosf"}, {"tan", "tanf"}, {"asin", "asinf"}, {"acos", "acosf"}, {"atan", "atanf"},
        {"atan2", "atan2f"}, {"sinh", "sinhf"}, {"cosh", "coshf"}, {"tanh", "tanhf"}, {"asinh", "asinhf"}, {"acosh", "acoshf"},
        {"atanh", "atanhf"}, {"sqrt", "sqrtf"}, {"inverseSqrt", "rsqrtf"}, {"exp", "expf"}, {"exp2", "exp2f"}, {"log", "logf"},
        {"log2", "log2f"}, {"floor", "floorf"}, {"ceil", "ceilf"}, {"round", "rintf"}, {"trunc", "truncf"}, {"fract", "mcx_fract"},
        {"sign", "mcx_sign"}, {"min", "fminf"}, {"max", "fmaxf"}, {"clamp", "mcx_clamp"}, {"mix", "mcx_mix"}, {"step", "mcx_step"},
        {"smoothstep", "mcx_smoothstep"}, {"pow", "powf"}, {"fma", "fmaf"}, {"saturate", "__saturatef"}, {"degrees", "mcx_degrees"},
        {"radians", "mcx_radians"}};
    if (math >= 1) {
        const char* const fast[][2] = {{"sin", "mcx_sin"}, {"cos", "mcx_cos"}, {"tan", "mcx_tan"}, {"sinh", "mcx_sinh"}, {"cosh", "mcx_cosh"},
                                       {"pow", "mcx_pow"}, {"exp", "__expf"}, {"exp2", "__builtin_amdgcn_exp2f"}, {"log", "__logf"},
                                       {"log2", "__builtin_amdgcn_logf"}, {"sqrt", "__builtin_amdgcn_sqrtf"}};
        for (auto& kv : fast) b[kv[0]] = kv[1];
    }
    if (math >= 2) { b["sin"] = "__sinf"; b["cos"] = "__cosf"; b["tan"] = "__tanf"; }
    return b;
}

const std::vector<std::vector<std::string>> kLevels = {{"||"}, {"&&"}, {"|"}, {"^"}, {"&"}, {"==", "!="}, {"<", ">", "<=", ">="},
                                                       {"<<", ">>"}, {"+", "-"}, {"*", "/", "%"}};

std::string join(const std::vector<std::string>& v, const char* sep) {
    std::string s;
    for (size_t i = 0; i < v.size(); ++i) { if (i) s += sep; s += v[i]; }
    return s;
}

struct Parser {
    std::vector<Token> toks;
    size_t i = 0;
    int slot;
    std::map<std::string, std::string> builtins;
    std::vector<std::string> local_functions;

    const Token& peek(size_t k = 0) const { static const Token end{END, "", 0}; return i + k < toks.size() ? toks[i + k] : end; }
    Token take() { Token t = peek(); ++i; return t; }
    bool is(Kind k, const char* v, size_t ahead = 0) const { const Token& t = peek(ahead); return t.kind == k && t.text == v; }
    bool accept(const char* v) { if (peek().text == v && peek().kind != NUM && peek().kind != END) { ++i; return true; } return false; }
    void expect(const char* v) { if (!accept(v)) bad(std::string("WGSL function string: expected '") + v + "' but found '" + peek().text + "'"); }
    std::string ident() { Token t = take(); if (t.kind != ID) bad("WGSL function string: expected an identifier, found '" + t.text + "'"); return t.text; }
    std::string type_name() {
        const std::string name = ident();
        auto it = kTypes.find(name);
        if (it == kTypes.end()) bad("WGSL function string: unsupported type '" + name + "' (scalar f32/i32/u32/bool only)");
        return it->second;
    }
    std::string fn_name(const std::string& name) const { return "mcx_uf" + std::to_string(slot) + "_" + name; }
    // a WGSL variable may be called mcx_something: keep it out of the device library's namespace, at every mention
    static std::string var_name(const std::string& name) { return name.rfind("mcx_", 0) == 0 ? name + "_v" : name; }
    bool is_local(const std::string& name) const { for (auto& f : local_functions) if (f == name) return true; return false; }

    // text from a caller: the recursive descent is bounded (parentheses, unary operators and blocks nest at most kMaxNesting deep)
    int nesting = 0;
    struct Nest {
        int& n;
        explicit Nest(int& depth) : n(depth) { if (++n > 200) bad("WGSL function string: nesting deeper than 200 levels"); }
        ~Nest() { --n; }
    };

    std::string expression(size_t level = 0) {
        if (level == kLevels.size()) return unary();
        std::string left = expression(level + 1);
        for (;;) {
            const Token& t = peek();
            bool hit = false;
            if (t.kind == OP) for (auto& op : kLevels[level]) if (t.text == op) hit = true;
            if (!hit) break;
            const std::string op = take().text;
            const std::string right = expression(level + 1);
            left = op == "%" ? "mcx_mod(" + left + ", " + right + ")" : "(" + left + " " + op + " " + right + ")";
        }
        return left;
    }
    std::string unary() {
        Nest guard(nesting);
        for (const char* op : {"-", "!", "~"})
            if (is(OP, op)) { take(); return std::string("(") + op + unary() + ")"; }
        return primary();
    }
    std::vector<std::string> call_args() {
        std::vector<std::string> args;
        if (!accept(")")) {
            for (;;) {
                args.push_back(expression());
                if (accept(")")) break;
                expect(",");
            }
        }
        return args;
    }
    std::string primary() {
        const Token t = take();
        if (t.kind == NUM) return number(t.text);
        if (t.kind == OP && t.text == "(") {
            const std::string inner = expression();
            expect(")");
            return "(" + inner + ")";
        }
        if (t.kind != ID) bad("WGSL function string: unexpected token '" + t.text + "'");
        const std::string& value = t.text;
        if (value == "true" || value == "false") return value;
        if (accept("(")) {
            const std::vector<std::string> args = call_args();
            const std::string list = join(args, ", ");
            auto ty = kTypes.find(value);
            if (ty != kTypes.end()) {
                if (args.size() != 1) bad("WGSL function string: " + value + "() takes one argument");
                return "((" + ty->second + ")(" + args[0] + "))";
            }
            if (value == "select") {
                if (args.size() != 3) bad("WGSL function string: select() takes three arguments");
                return "((" + args[2] + ") ? (" + args[1] + ") : (" + args[0] + "))";
            }
            int n = 0;
            if (value == "pow" && args.size() == 2 && whole_exponent(args[1], &n)) {
                const std::string chain = "McxPowI<" + std::to_string(abs(n)) + ">::of(" + args[0] + ")";
                return n >= 0 ? chain : "(1.0f / " + chain + ")";
            }
            auto b = builtins.find(value);
            if (b != builtins.end()) return b->second + "(" + list + ")";
            auto tc = kTableCalls.find(value);
            if (tc != kTableCalls.end() && !is_local(value)) return tc->second + "(" + list + ")";
            if (is_local(value)) return fn_name(value) + "(" + list + ")";
            if (value.rfind("vec", 0) == 0 || value.rfind("mat", 0) == 0 || value == "array")
                bad("WGSL function string: '" + value + "' is not supported (scalar code only)");
            return fn_name(value) + "(" + list + ")";                       // a helper defined later in the same string
        }
        if (is(OP, ".") || is(OP, "[")) bad("WGSL function string: member / index access is not supported (scalar code only)");
        return var_name(value);
    }

    static std::string pad(int indent) { return std::string(4 * (size_t)indent, ' '); }
    std::vector<std::string> block(int indent) {
        Nest guard(nesting);
        expect("{");
        std::vector<std::string> out;
        while (!accept("}")) {
            if (peek().kind == END) bad("WGSL function string: unbalanced braces");
            for (auto& line : statement(indent)) out.push_back(line);
        }
        return out;
    }
    static bool plain_int(const std::string& s) {                             // \(?-?\d+\)?
        size_t a = 0, b = s.size();
        if (a < b && s[a] == '(') ++a;
        if (a < b && s[b - 1] == ')') --b;
        if (a < b && s[a] == '-') ++a;
        if (a >= b) return false;
        for (size_t j = a; j < b; ++j) if (!isdigit((unsigned char)s[j])) return false;
        return true;
    }
    std::string simple_statement() {
        const Token t = peek();
        if (t.kind == ID && (t.text == "let" || t.text == "var" || t.text == "const")) {
            take();
            const std::string name = var_name(ident());
            std::string ctype = "auto";
            if (accept(":")) ctype = type_name();
            if (accept("=")) {
                const std::string init = expression();
                if (ctype == "auto" && plain_int(init)) ctype = "int";
                return std::string(t.text == "const" ? "const " : "") + ctype + " " + name + " = " + init;
            }
            if (ctype == "auto") bad("WGSL function string: a declaration needs a type or an initialiser");
            return ctype + " " + name + " = 0";
        }
        const std::string called = ident();
        if (accept("(")) {
            const std::vector<std::string> args = call_args();
            auto b = builtins.find(called);
            return (b != builtins.end() ? b->second : fn_name(called)) + "(" + join(args, ", ") + ")";
        }
        const std::string target = var_name(called);
        const std::string op = take().text;
        if (op == "++" || op == "--") return target + op;
        if (op == "=" || op == "+=" || op == "-=" || op == "*=" || op == "/=" || op == "%=" || op == "&=" || op == "|=" || op == "^=" ||
            op == "<<=" || op == ">>=") {
            const std::string value = expression();
            if (op == "%=") return target + " = mcx_mod(" + target + ", " + value + ")";
            return target + " " + op + " " + value;
        }
        bad("WGSL function string: unsupported statement near '" + target + " " + op + "'");
    }
    std::vector<std::string> statement(int indent) {
        const std::string p = pad(indent);
        const Token t = peek();
        std::vector<std::string> out;
        auto append = [&](const std::vector<std::string>& lines) { for (auto& l : lines) out.push_back(l); };
        if (t.kind == OP && t.text == "{") { out.push_back(p + "{"); append(block(indent + 1)); out.push_back(p + "}"); return out; }
        if (t.kind == OP && t.text == ";") { take(); return out; }
        if (t.kind == ID && t.text == "return") {
            take();
            if (accept(";")) return {p + "return 0.0f;"};
            const std::string e = expression();
            expect(";");
            return {p + "return mcx_b2f(" + e + ");"};
        }
        if (t.kind == ID && t.text == "if") {
            take();
            std::string cond = expression();
            out.push_back(p + "if (" + cond + ") {");
            append(block(indent + 1));
            while (is(ID, "else")) {
                take();
                if (is(ID, "if")) {
                    take();
                    cond = expression();
                    out.push_back(p + "} else if (" + cond + ") {");
                    append(block(indent + 1));
                } else {
                    out.push_back(p + "} else {");
                    append(block(indent + 1));
                    break;
                }
            }
            out.push_back(p + "}");
            return out;
        }
        if (t.kind == ID && t.text == "while") {
            take();
            const std::string cond = expression();
            out.push_back(p + "while (" + cond + ") {"); append(block(indent + 1)); out.push_back(p + "}");
            return out;
        }
        if (t.kind == ID && t.text == "loop") {
            take();
            out.push_back(p + "while (true) {"); append(block(indent + 1)); out.push_back(p + "}");
            return out;
        }
        if (t.kind == ID && t.text == "for") {
            take();
            expect("(");
            const std::string init = is(OP, ";") ? "" : simple_statement();
            expect(";");
            const std::string cond = is(OP, ";") ? "" : expression();
            expect(";");
            const std::string step = is(OP, ")") ? "" : simple_statement();
            expect(")");
            out.push_back(p + "for (" + init + "; " + cond + "; " + step + ") {"); append(block(indent + 1)); out.push_back(p + "}");
            return out;
        }
        if (t.kind == ID && (t.text == "break" || t.text == "continue")) { take(); expect(";"); return {p + t.text + ";"}; }
        const std::string text = simple_statement();
        expect(";");
        return {p + text + ";"};
    }
    // one function; returns its HIP text
    std::string function(const std::string& emitted_name) {
        while (accept("@")) {                                               // attributes such as @must_use
            ident();
            if (accept("(")) call_args();
        }
        if (!is(ID, "fn")) bad("WGSL function string must start with 'fn'");
        take();
        const std::string original = ident();
        expect("(");
        std::vector<std::string> params;
        if (!accept(")")) {
            for (;;) {
                const std::string pname = var_name(ident());
                expect(":");
                params.push_back(type_name() + " " + pname);
                if (accept(")")) break;
                expect(",");
            }
        }
        std::string rtype = "float";
        if (accept("->")) rtype = type_name();
        local_functions.push_back(original);
        std::vector<std::string> body = block(1);
        body.push_back(rtype != "void" ? "    return 0;" : "");
        return "MCX_DEV " + rtype + " " + emitted_name + "(" + join(params, ", ") + ") {\n" + join(body, "\n") + "\n}";
    }
};

std::string rstrip(std::string s) { while (!s.empty() && isspace((unsigned char)s.back())) s.pop_back(); return s; }

std::string translate(const std::string& wgsl, int slot, const std::string& entry_name, int math) {
    Parser ps;
    ps.toks = tokenize(wgsl);
    if (ps.toks.empty()) bad("empty WGSL function string");
    ps.slot = slot;
    ps.builtins = builtins_for(math);
    std::vector<std::string> names;                    // pre-scan helper names so that calls are prefixed consistently
    for (size_t j = 0; j + 1 < ps.toks.size(); ++j)
        if (ps.toks[j].kind == ID && ps.toks[j].text == "fn" && ps.toks[j + 1].kind == ID) names.push_back(ps.toks[j + 1].text);
    if (names.empty()) bad("WGSL function string must contain a function definition ('fn name(...)')");
    ps.local_functions = names;
    std::vector<std::string> pieces, declarations;
    while (ps.peek().kind != END) {
        if (ps.is(OP, ";")) { ps.take(); continue; }
        if (pieces.size() >= names.size()) bad("WGSL function string must start with 'fn'");
        const std::string text = ps.function(pieces.empty() ? entry_name : ps.fn_name(names[pieces.size()]));
        declarations.push_back(rstrip(text.substr(0, text.find('{'))) + ";");
        pieces.push_back(text);
    }
    const std::string entry_calls = ps.fn_name(names[0]);
    std::string alias;                                 // helpers may call the entry by its WGSL name: provide the alias
    bool called = false;
    for (auto& p : pieces) if (p.find(entry_calls + "(") != std::string::npos) called = true;
    if (called) {
        const std::string sig = declarations[0].substr(0, declarations[0].size() - 1);
        const size_t lp = sig.find('('), rp = sig.rfind(')');
        const std::string params = sig.substr(lp + 1, rp - lp - 1);
        std::vector<std::string> arg_names;
        size_t a = 0;
        while (a <= params.size()) {
            size_t b = params.find(',', a);
            if (b == std::string::npos) b = params.size();
            std::string one = params.substr(a, b - a);
            const size_t e = one.find_last_not_of(" \t");
            if (e != std::string::npos) {
                one = one.substr(0, e + 1);
                const size_t sp = one.find_last_of(" \t");
                arg_names.push_back(sp == std::string::npos ? one : one.substr(sp + 1));
            }
            a = b + 1;
        }
        // "MCX_DEV <rtype> name(": the second word, or the second and third for "unsigned int"
        const size_t w1 = sig.find(' ');
        const size_t name_at = sig.rfind(' ', lp);
        const std::string rtype = sig.substr(w1 + 1, name_at - w1 - 1);
        alias = "MCX_DEV " + rtype + " " + entry_calls + "(" + params + ") { return " + entry_name + "(" + join(arg_names, ", ") + "); }\n";
        declarations.push_back("MCX_DEV " + rtype + " " + entry_calls + "(" + params + ");");
    }
    return join(declarations, "\n") + "\n" + join(pieces, "\n") + "\n" + alias;
}

const char kPrelude[] =
    "\ntemplate <int N> struct McxPowI {\n"
    "    static MCX_DEV float of(float x) {\n"
    "        if constexpr (N == 0) return 1.0f;\n"
    "        else if constexpr (N == 1) return x;\n"
    "        else if constexpr (N % 2 == 0) { float h = McxPowI<N / 2>::of(x); return h * h; }\n"
    "        else return McxPowI<N - 1>::of(x) * x;\n"
    "    }\n"
    "};\n";


// ---- planning of a whole payload (include/mcx.h: mcx_wgsl_plan) --------------------------------------------------------------
// What the reference's Python half hands its native module is K strings; what they ARE is visible in their text. The three
// recognitions below are the ones wgpu_montecarlo/_core.py made in Python until the binding's planning moved here
// (tests/core_reference_planner.py keeps that Python as the test-suite's independent restatement).

bool tok_is(const std::vector<Token>& t, size_t j, Kind k, const std::string& text) { return j < t.size() && t[j].kind == k && t[j].text == text; }

// consumes `expected` (space-separated token texts) at position j; "%ID" matches any identifier, "%NUM" any number
bool match_seq(const std::vector<Token>& t, size_t* j, const char* expected) {
    size_t at = *j;
    std::string word;
    for (const char* c = expected;; ++c) {
        if (*c == ' ' || *c == '\0') {
            if (!word.empty()) {
                if (at >= t.size()) return false;
                if (word == "%ID") { if (t[at].kind != ID) return false; }
                else if (word == "%NUM") { if (t[at].kind != NUM) return false; }
                else if (t[at].text != word || t[at].kind == END) return false;
                ++at;
                word.clear();
            }
            if (*c == '\0') break;
        } else word += *c;
    }
    *j = at;
    return true;
}

std::string replace_all(std::string s, const std::string& from, const std::string& to) {
    for (size_t at = 0; (at = s.find(from, at)) != std::string::npos; at += to.size()) s.replace(at, from.size(), to);
    return s;
}
std::string strip(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

struct Weighted { std::vector<std::string> f_texts; bool p_analytic = false, q_analytic = false; std::string p_text, q_text; };

// Every string one of the reference's importance-sampling wrappers (python/wgpu_montecarlo/__init__.py:893-899, 968-974) around
// the same p and q:  fn _is_wrapper_i(x: f32) -> f32 { let f_val = _is_f_orig_i(x); let p = <P>(x); let q = <Q>(x); return f_val * p / q; }
// followed by the definitions of _is_pdf_p_i / _is_pdf_q_i (when analytic) and _is_f_orig_i (possibly with helpers).
bool split_weighted(const char* const* functions, int k, Weighted* out) {
    std::string p_seen, q_seen;
    bool first = true;
    for (int n = 0; n < k; ++n) {
        const std::string text = functions[n] ? functions[n] : "";
        std::vector<Token> t;
        try { t = tokenize(text); } catch (const TranslateError&) { return false; }
        if (t.size() < 4 || !tok_is(t, 0, ID, "fn") || t[1].kind != ID || t[1].text.rfind("_is_wrapper_", 0) != 0) return false;
        const std::string i = t[1].text.substr(strlen("_is_wrapper_"));
        if (i.empty()) return false;
        for (char c : i) if (!isdigit((unsigned char)c)) return false;
        size_t j = 2;
        if (!match_seq(t, &j, "( x : f32 ) -> f32 { let f_val =") || !tok_is(t, j, ID, "_is_f_orig_" + i)) return false;
        ++j;
        if (!match_seq(t, &j, "( x ) ; let p =") || j >= t.size()) return false;
        const std::string p_call = t[j++].text;
        if (!match_seq(t, &j, "( x ) ; let q =") || j >= t.size()) return false;
        const std::string q_call = t[j++].text;
        if (!match_seq(t, &j, "( x ) ; return f_val * p / q ; }")) return false;
        const bool p_tab = p_call == "pdf_target_from_table", q_tab = q_call == "pdf_proposal_from_table";
        if (!p_tab && p_call != "_is_pdf_p_" + i) return false;
        if (!q_tab && q_call != "_is_pdf_q_" + i) return false;
        // the definitions that follow: each top-level `fn name(` starts one; helpers stay with the named one they follow
        struct Part { std::string name; size_t begin; };
        std::vector<Part> starts;
        int depth = 0;
        const size_t rest_tok = j;
        for (size_t a = j; a < t.size(); ++a) {
            if (t[a].kind == OP && t[a].text == "{") ++depth;
            else if (t[a].kind == OP && t[a].text == "}") --depth;
            else if (depth == 0 && t[a].kind == ID && t[a].text == "fn" && a + 2 < t.size() && t[a + 1].kind == ID && tok_is(t, a + 2, OP, "("))
                starts.push_back({t[a + 1].text, t[a].pos});
        }
        if (starts.empty() || t[rest_tok].pos != starts[0].begin) return false;            // something before the first definition
        std::string parts[3];                                                            // p, q, f
        const std::string names[3] = {"_is_pdf_p_" + i, "_is_pdf_q_" + i, "_is_f_orig_" + i};
        int current = -1;
        for (size_t a = 0; a < starts.size(); ++a) {
            const size_t end = a + 1 < starts.size() ? starts[a + 1].begin : text.size();
            int which = -1;
            for (int w = 0; w < 3; ++w) if (starts[a].name == names[w]) which = w;
            if (which >= 0) current = which;
            else if (current < 0) return false;
            parts[current] += text.substr(starts[a].begin, end - starts[a].begin);
        }
        if (parts[2].empty() || parts[0].empty() == !p_tab || parts[1].empty() == !q_tab) return false;
        const std::string p_norm = p_tab ? "" : strip(replace_all(parts[0], names[0], "_is_pdf_p"));
        const std::string q_norm = q_tab ? "" : strip(replace_all(parts[1], names[1], "_is_pdf_q"));
        if (first) { p_seen = p_norm; q_seen = q_norm; out->p_analytic = !p_tab; out->q_analytic = !q_tab; first = false; }
        else if (p_norm != p_seen || q_norm != q_seen || out->p_analytic != !p_tab || out->q_analytic != !q_tab) return false;
        out->f_texts.push_back(parts[2]);
    }
    out->p_text = p_seen;
    out->q_text = q_seen;
    return k > 0;
}

bool literal_value(const std::vector<Token>& t, size_t* j, double* v) {               // [-|+] NUM
    double sign = 1.0;
    if (tok_is(t, *j, OP, "-")) { sign = -1.0; ++*j; } else if (tok_is(t, *j, OP, "+")) ++*j;
    if (*j >= t.size() || t[*j].kind != NUM) return false;
    char* end = nullptr;
    *v = sign * strtod(t[*j].text.c_str(), &end);
    ++*j;
    return true;
}

// Is `text` the closure Distribution.normal(mean, std) hands the transpiler (python/wgpu_montecarlo/__init__.py:343-347:
// exp(-0.5 z z) / (sigma sqrt_2pi), z = (x - mean) / sigma) for exactly the parameters the call samples with?
bool is_normal_pdf_text(const std::string& text, float mean, float std_) {
    std::vector<Token> t;
    try { t = tokenize(text); } catch (const TranslateError&) { return false; }
    size_t j = 0;
    if (!match_seq(t, &j, "fn %ID ( x : f32 ) -> f32 {")) return false;
    double c_mean = 0, c_sigma = 0, c_s2pi = 0;
    int seen = 0;
    while (tok_is(t, j, ID, "const")) {
        ++j;
        if (j >= t.size() || t[j].kind != ID) return false;
        const std::string name = t[j++].text;
        double v = 0;
        if (!match_seq(t, &j, ": f32 =") || !literal_value(t, &j, &v) || !match_seq(t, &j, ";")) return false;
        if (name == "mean") { c_mean = v; seen |= 1; } else if (name == "sigma") { c_sigma = v; seen |= 2; }
        else if (name == "sqrt_2pi") { c_s2pi = v; seen |= 4; } else return false;
    }
    if (seen != 7 || c_mean != (double)mean || c_sigma != (double)std_ || fabs(c_s2pi - 2.5066282746310002) >= 1e-12) return false;
    return match_seq(t, &j, "var z = ( ( x - mean ) / sigma ) ; return ( exp ( ( ( ( - 0.5 ) * z ) * z ) ) / ( sigma * sqrt_2pi ) ) ; }") && j == t.size();
}

// exactly the transpiler's text for x, x**2, .., x**K (K >= 8): `return x;`, `return pow(x, k.0);`
bool moment_family(const char* const* functions, int k) {
    if (k < 8) return false;
    for (int n = 0; n < k; ++n) {
        std::vector<Token> t;
        try { t = tokenize(functions[n] ? functions[n] : ""); } catch (const TranslateError&) { return false; }
        size_t j = 0;
        if (!match_seq(t, &j, "fn %ID ( x : f32 ) -> f32 { return")) return false;
        if (n == 0) { if (!match_seq(t, &j, "x ; }") || j != t.size()) return false; continue; }
        if (!match_seq(t, &j, "pow ( x , %NUM ) ; }") || j != t.size()) return false;
        int e = 0;
        if (!whole_exponent(number(t[j - 4].text), &e) || e != n + 1) return false;
    }
    return true;
}

std::string f32_literal(float v) {
    char buf[48];
    snprintf(buf, sizeof buf, "%.9g", (double)v);
    std::string s = buf;
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";                      // "2" -> "2.0" (inf / nan never reach here)
    return s + "f";
}

// HIP text of the reference's analytic log-density of one distribution type, generate_log_pdf_code_for_dist
// (src/shader_gen.rs:543-571): what its MH step evaluates when integrate_mcmc gets no table. The normal case is `pow(z, 2.0)` in
// the reference's WGSL -- backend-defined for z < 0 -- and is emitted as the intended z * z.
bool analytic_logpdf(const char* name, int dist_type, float p1, float p2, std::string* out) {
    const std::string a = f32_literal(p1), b = f32_literal(p2);
    std::string body;
    if (dist_type == MCX_DIST_UNIFORM) body = "((" + a + " <= x) && (x < " + b + ")) ? -logf(" + b + " - " + a + ") : -100.0f";
    else if (dist_type == MCX_DIST_NORMAL)
        body = "-0.5f * (((x - " + a + ") / " + b + ") * ((x - " + a + ") / " + b + ")) - logf(" + b + " * 2.50662827463f)";
    else if (dist_type == MCX_DIST_EXPONENTIAL) body = "(x >= 0.0f) ? logf(" + a + ") - " + a + " * x : -100.0f";
    else return false;
    *out = std::string("MCX_DEV float ") + name + "(float x) { return " + body + "; }";
    return true;
}

int plan(const mcx_wgsl_program& g, mcx_module_desc* d, std::string* src) {
    const bool literal = g.math == 0;
    std::vector<std::string> parts = {kPrelude};
    auto functions_as_given = [&] {
        for (int i = 0; i < g.k; ++i) parts.push_back(translate(g.functions[i] ? g.functions[i] : "", i, "user_func_" + std::to_string(i), g.math));
    };
    d->kind = g.kind; d->k = g.k; d->dist_type = g.dist_type;
    if (g.kind == MCX_KIND_INTEGRATE) {
        Weighted w;
        bool split = !literal && split_weighted(g.functions, g.k, &w);
        if (split && (w.p_analytic == (g.have_target_table != 0) || w.q_analytic == (g.have_proposal_table != 0))) split = false;   // a wrapper reads a table the call did not bring (or the reverse): literal
        if (split) {
            // the reference's importance-sampling call: K integrands + ONE weight p / q per sample instead of K evaluations of its text
            const bool q_sampler = w.q_analytic && g.dist_type == MCX_DIST_NORMAL && is_normal_pdf_text(w.q_text, g.param1, g.param2);
            for (int i = 0; i < g.k; ++i) parts.push_back(translate(w.f_texts[i], i, "user_func_" + std::to_string(i), g.math));
            if (w.p_analytic) parts.push_back(translate(w.p_text, g.k, "mcx_pdf_p", g.math));
            if (w.q_analytic && !q_sampler) parts.push_back(translate(w.q_text, g.k + 1, "mcx_pdf_q", g.math));
            d->weight = 1; d->p_table = w.p_analytic ? 0 : 1; d->q_table = w.q_analytic ? 0 : 1; d->q_sampler = q_sampler ? 1 : 0;
        } else {
            functions_as_given();
            d->user_tables = (g.have_target_table ? 1 : 0) | (g.have_proposal_table ? 2 : 0);
            d->moment_family = (!literal && d->user_tables == 0 && g.k <= 32 && moment_family(g.functions, g.k)) ? 1 : 0;
        }
    } else if (g.kind == MCX_KIND_MCMC) {
        functions_as_given();
        // the log-PDF tables are optional (src/lib.rs:296-304): without one, the MH step evaluates the analytic log-density of that
        // distribution type (src/shader_gen.rs:327-339, 496-509). A normal proposal's log q is -z^2/2 + const of the deviate the
        // sampler holds (not with math = precise): neither its table nor its text is used.
        std::string text;
        if (!g.have_target_table) {
            if (!analytic_logpdf("mcx_logpdf_p", g.target_dist_type, g.target_param1, g.target_param2, &text))
                return mcx::fail(MCX_E_RUNTIME, "Failed to create MCMC pipeline: a custom distribution needs its log-PDF table");
            parts.push_back(text);
            d->logpdf_analytic |= 1;
        }
        d->q_sampler = (!literal && g.dist_type == MCX_DIST_NORMAL) ? 1 : 0;
        if (!d->q_sampler && !g.have_proposal_table) {
            if (!analytic_logpdf("mcx_logpdf_q", g.dist_type, g.param1, g.param2, &text))
                return mcx::fail(MCX_E_RUNTIME, "Failed to create MCMC pipeline: a custom distribution needs its log-PDF table");
            parts.push_back(text);
            d->logpdf_analytic |= 2;
        }
    } else return mcx::fail(MCX_E_INVALID, "mcx_wgsl_plan: kind must be MCX_KIND_INTEGRATE or MCX_KIND_MCMC");
    *src = join(parts, "\n\n");
    return MCX_OK;
}

}  // namespace

extern "C" {

const char* mcx_wgsl_prelude(void) { return kPrelude; }

int mcx_wgsl_translate(const char* wgsl, int32_t slot, const char* entry_name, int32_t math, char** out_text) {
    if (!wgsl || !entry_name || !out_text) return mcx::fail(MCX_E_INVALID, "mcx_wgsl_translate: null argument");
    if (math < 0 || math > 2) return mcx::fail(MCX_E_INVALID, "math must be one of ('precise', 'default', 'fast')");
    try {
        const std::string text = translate(wgsl, slot, entry_name, math);
        char* buf = (char*)malloc(text.size() + 1);
        if (!buf) return mcx::fail(MCX_E_RUNTIME, "out of memory");
        memcpy(buf, text.c_str(), text.size() + 1);
        *out_text = buf;
        return MCX_OK;
    } catch (const TranslateError& e) {
        return mcx::fail(MCX_E_TRANSLATE, e.msg);
    }
}

int mcx_wgsl_plan(const mcx_wgsl_program* prog, mcx_module_desc* desc_out, char** user_src_out) {
    if (!prog || !desc_out || !user_src_out) return mcx::fail(MCX_E_INVALID, "mcx_wgsl_plan: null argument");
    if (prog->struct_size != sizeof(mcx_wgsl_program)) return mcx::fail(MCX_E_INVALID, "mcx_wgsl_plan: program of another ABI version (mcx_wgsl_program_init)");
    if (prog->k <= 0 || !prog->functions) return mcx::fail(MCX_E_INVALID, "At least one function is required");
    if (prog->math < 0 || prog->math > 2) return mcx::fail(MCX_E_INVALID, "math must be one of ('precise', 'default', 'fast')");
    mcx_module_desc_init(desc_out);
    try {
        std::string src;
        if (int rc = plan(*prog, desc_out, &src)) return rc;
        char* buf = (char*)malloc(src.size() + 1);
        if (!buf) return mcx::fail(MCX_E_RUNTIME, "out of memory");
        memcpy(buf, src.c_str(), src.size() + 1);
        *user_src_out = buf;
        return MCX_OK;
    } catch (const TranslateError& e) {
        return mcx::fail(MCX_E_TRANSLATE, e.msg);
    }
}

}  // extern "C"
