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
struct Token { Kind kind; std::string text; };

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
            out.push_back({NUM, text.substr(start, i - start)});
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
            out.push_back({NUM, text.substr(start, i - start)});
            continue;
        }
        if (isalpha((unsigned char)text[i]) || text[i] == '_') {
            while (i < n && (isalnum((unsigned char)text[i]) || text[i] == '_')) ++i;
            out.push_back({ID, text.substr(start, i - start)});
            continue;
        }
        bool matched = false;
        for (const char* op : kOps3) if (text.compare(i, 3, op) == 0) { out.push_back({OP, op}); i += 3; matched = true; break; }
        if (matched) continue;
        for (const char* op : kOps2) if (text.compare(i, 2, op) == 0) { out.push_back({OP, op}); i += 2; matched = true; break; }
        if (matched) continue;
        if (strchr(kOps1, text[i]) && text[i] != '\0') { out.push_back({OP, std::string(1, text[i])}); ++i; continue; }
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

    const Token& peek(size_t k = 0) const { static const Token end{END, ""}; return i + k < toks.size() ? toks[i + k] : end; }
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

}  // extern "C"
