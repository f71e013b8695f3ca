// Device audio front-end: waveform -> (resample to 22 kHz) -> MFCC(64) as the reference computes it on the CPU with
// torchaudio (data_utils/utils.py:148-231: Resample(sr_0, 22000) per channel, MFCC(n_mfcc=64, n_fft=2048, n_mels=256,
// hop=734|1467, mel_scale='htk')).  torchaudio is third-party and not available in this image: the constants below
// restate its published definitions (sinc_interp_hann kernel with lowpass_filter_width 6 / rolloff 0.99; periodic Hann;
// HTK mel filterbank f_min 0, f_max sr/2, norm None; 10*log10(clamp 1e-10), top_db 80 per clip; orthonormal DCT-II).
// PARITY UNPINNED against torchaudio; pinned against talkshow_amd/frontend.py (same formulae in numpy) on the GPU.
// The STFT is one kernel (framing + window + a radix-4 Stockham real FFT in LDS + |X|^2, mfcc.hip); the mel projection and the
// DCT are GEMMs on conv_gemm_f32.  (Rounds 1-3 ran the transform as a 2048 x 2050 DFT matrix: 1.26 GMAC per 10 s clip.)
#include <cmath>
#include <mutex>

#include "host_common.h"

using namespace ts;

namespace ts {
hipError_t launch_resample_polyphase(const float *x, int B, int N, const float *kern, int norig, int nnew, int width, int kw,
                                     float *out, int Nout, hipStream_t s);
hipError_t launch_stft_power(const float *x, int B, int N, int T, int hop, const float *win, const float *tw1024, const float *tw2048,
                             float *pw, int ldp, hipStream_t s);
hipError_t launch_resample_kaiser(const float *x, int B, int N, const float *win, const float *delta, int nwin, int num_table,
                                  double ratio, float *out, int Nout, int ldo, hipStream_t s);
hipError_t launch_db_topdb(float *mel, int B, long per_clip, float top_db, hipStream_t s);
}  // namespace ts

struct ts_mfcc {
    ts_ctx *ctx = nullptr;
    int sr_in = 0, sr_out = 0, norig = 1, nnew = 1, width = 0, kw = 0;
    int nfft = 2048, hop = 734, nmels = 256, nmfcc = 64, nbins = 1025, nbins_pad = 1056;
    DevBuf rs_kern, window, tw1024, tw2048;   // FFT twiddles: exp(-2 pi i m / 1024), m < 1024; exp(-2 pi i k / 2048), k <= 1024
    ConvLayer mel, dct;
    struct Work {
        DevBuf x22, power, melb;
    };
    StreamWorks<Work> works;
    Work &work(hipStream_t s) { return works.get(s); }
    long resampled_len(long N) const { return (nnew * N + norig - 1) / norig; }
};

static int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

extern "C" {

int ts_mfcc_create(ts_ctx *ctx, int sr_in, int sr_out, int fps, ts_mfcc **out) {
    if (!ctx || !out) return fail("ts_mfcc_create: null argument");
    if (fps != 30 && fps != 15) return fail("ts_mfcc_create: fps must be 15 or 30 (hop 1467 / 734, utils.py:157-160)");
    TS_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<ts_mfcc> m(new ts_mfcc());
    m->ctx = ctx;
    m->sr_in = sr_in;
    m->sr_out = sr_out;
    m->hop = fps == 30 ? 734 : 1467;
    const double PI = 3.14159265358979323846;
    // ---- resampling kernel (torchaudio _get_sinc_resample_kernel, sinc_interp_hann) ----
    if (sr_in != sr_out) {
        const int g = gcd_i(sr_in, sr_out);
        m->norig = sr_in / g;
        m->nnew = sr_out / g;
        const double lpw = 6.0, rolloff = 0.99;
        const double base = std::min(m->norig, m->nnew) * rolloff;
        m->width = (int)std::ceil(lpw * m->norig / base);
        m->kw = 2 * m->width + m->norig;
        std::vector<float> k((size_t)m->nnew * m->kw);
        for (int ph = 0; ph < m->nnew; ++ph)
            for (int i = 0; i < m->kw; ++i) {
                double t = ((double)(-ph) / m->nnew + (double)(i - m->width) / m->norig) * base;
                t = std::max(-lpw, std::min(lpw, t));
                const double win = std::pow(std::cos(t * PI / lpw / 2.0), 2.0);
                const double tp = t * PI;
                const double sinc = tp == 0.0 ? 1.0 : std::sin(tp) / tp;
                k[(size_t)ph * m->kw + i] = (float)(sinc * win * (base / m->norig));
            }
        TS_TRY(m->rs_kern.upload(k.data(), k.size() * sizeof(float)));
    }
    // ---- periodic Hann window ----
    {
        std::vector<float> w(m->nfft);
        for (int n = 0; n < m->nfft; ++n) w[n] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * n / m->nfft));
        TS_TRY(m->window.upload(w.data(), w.size() * sizeof(float)));
    }
    // ---- FFT twiddles, computed in double: the 1024-point complex passes and the real-transform split ----
    {
        if (m->nfft != 2048) return fail("ts_mfcc_create: the STFT kernel is built for n_fft = 2048");
        std::vector<float> a(2 * 1024), bq(2 * 1025);
        for (int i = 0; i < 1024; ++i) {
            a[2 * i] = (float)std::cos(2.0 * PI * i / 1024.0);
            a[2 * i + 1] = (float)(-std::sin(2.0 * PI * i / 1024.0));
        }
        for (int i = 0; i <= 1024; ++i) {
            bq[2 * i] = (float)std::cos(2.0 * PI * i / 2048.0);
            bq[2 * i + 1] = (float)(-std::sin(2.0 * PI * i / 2048.0));
        }
        TS_TRY(m->tw1024.upload(a.data(), a.size() * sizeof(float)));
        TS_TRY(m->tw2048.upload(bq.data(), bq.size() * sizeof(float)));
    }
    // ---- HTK mel filterbank (torchaudio.functional.melscale_fbanks, norm=None), transposed to [n_mels][n_freqs] ----
    {
        const int nb = m->nbins, nm = m->nmels;
        auto hz2mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
        auto mel2hz = [](double x) { return 700.0 * (std::pow(10.0, x / 2595.0) - 1.0); };
        const double fmax = (double)(sr_out / 2);
        std::vector<double> fpts(nm + 2);
        const double m0 = hz2mel(0.0), m1 = hz2mel(fmax);
        for (int i = 0; i < nm + 2; ++i) fpts[i] = mel2hz(m0 + (m1 - m0) * i / (nm + 1));
        std::vector<float> fb((size_t)nm * m->nbins_pad, 0.f);
        for (int f = 0; f < nb; ++f) {
            const double freq = (double)(sr_out / 2) * f / (nb - 1);
            for (int j = 0; j < nm; ++j) {
                const double down = (freq - fpts[j]) / (fpts[j + 1] - fpts[j]);
                const double up = (fpts[j + 2] - freq) / (fpts[j + 2] - fpts[j + 1]);
                const double v = std::max(0.0, std::min(down, up));
                fb[(size_t)j * m->nbins_pad + f] = (float)v;
            }
        }
        TS_TRY(pack_linear_layer(fb.data(), m->nbins_pad, nullptr, nm, m->nbins_pad, &m->mel));
    }
    // ---- orthonormal DCT-II (torchaudio.functional.create_dct), [n_mfcc][n_mels] ----
    {
        const int nm = m->nmels, nc = m->nmfcc;
        std::vector<float> d((size_t)nc * nm);
        for (int k = 0; k < nc; ++k)
            for (int n = 0; n < nm; ++n) {
                double v = std::cos(PI / nm * (n + 0.5) * k);
                if (k == 0) v *= 1.0 / std::sqrt(2.0);
                d[(size_t)k * nm + n] = (float)(v * std::sqrt(2.0 / nm));
            }
        TS_TRY(pack_linear_layer(d.data(), nm, nullptr, nc, nm, &m->dct));
    }
    *out = m.release();
    return 0;
}
void ts_mfcc_destroy(ts_mfcc *m) { delete m; }

// stage 1 of ts_mfcc_forward on its own (get_mfcc_sepa resamples the whole clip first, then takes the MFCC of two parts):
// wav_dev (B,N) at sr_in -> out_dev (B, ts_mfcc_resampled_len(m, N)) at sr_out
long ts_mfcc_resampled_len(const ts_mfcc *m, long N) { return m ? m->resampled_len(N) : -1; }
int ts_mfcc_resample(ts_mfcc *m, const float *wav, int B, long N, float *out, void *stream) {
    if (!m || !wav || !out) return fail("ts_mfcc_resample: null argument");
    hipStream_t s = (hipStream_t)stream;
    MiscScope ms(m->ctx, s);
    if (m->sr_in == m->sr_out) {
        TS_HIP(hipMemcpyAsync(out, wav, (size_t)B * N * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    TS_HIP(launch_resample_polyphase(wav, B, (int)N, m->rs_kern.f(), m->norig, m->nnew, m->width, m->kw, out,
                                     (int)m->resampled_len(N), s));
    return 0;
}

// ---- librosa.load(sr=16000)'s resampler (face path, data_utils/utils.py:194): resampy 'kaiser_best' ---------------------
namespace {
struct KaiserTable {
    DevBuf win, delta;
    int nwin = 0, num_table = 512;
};
// half of a sinc windowed by a Kaiser window: 64 zero crossings, 512 samples per crossing (resampy's published
// 'kaiser_best' design: rolloff 0.9475937167399596, beta 14.769656459379492)
int kaiser_table(int device, const KaiserTable **out) {
    static std::mutex mu;
    static std::map<int, std::unique_ptr<KaiserTable>> tabs;
    std::lock_guard<std::mutex> g(mu);
    auto &t = tabs[device];
    if (!t) {
        const int num_zeros = 64, precision = 9, num_bits = 1 << precision, n = num_bits * num_zeros;
        const double rolloff = 0.9475937167399596, beta = 14.769656459379492, PI = 3.14159265358979323846;
        auto i0 = [](double x) {   // modified Bessel function of the first kind, order 0 (power series)
            double sum = 1.0, term = 1.0;
            for (int k = 1; k < 500; ++k) {
                term *= (x / (2.0 * k)) * (x / (2.0 * k));
                sum += term;
                if (term < 1e-18 * sum) break;
            }
            return sum;
        };
        std::vector<float> win(n + 1), delta(n + 1, 0.f);
        const int Mw = 2 * n + 1;   // the full symmetric Kaiser window; its right half [n:] tapers the sinc
        for (int i = 0; i <= n; ++i) {
            const double tpos = (double)num_zeros * i / n;                  // np.linspace(0, num_zeros, n + 1)
            const double a = rolloff * tpos * PI;
            const double sinc = a == 0.0 ? 1.0 : std::sin(a) / a;
            const double m = (double)(n + i) - (Mw - 1) / 2.0;            // scipy.signal.kaiser(M, beta)[n + i]
            const double r = 2.0 * m / (Mw - 1);
            const double taper = i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0(beta);
            win[i] = (float)(taper * rolloff * sinc);
        }
        for (int i = 0; i < n; ++i) delta[i] = win[i + 1] - win[i];
        std::unique_ptr<KaiserTable> k(new KaiserTable());
        k->nwin = n + 1;
        k->num_table = num_bits;
        TS_TRY(k->win.upload(win.data(), win.size() * sizeof(float)));
        TS_TRY(k->delta.upload(delta.data(), delta.size() * sizeof(float)));
        t = std::move(k);
    }
    *out = t.get();
    return 0;
}
}  // namespace

// output length of librosa.resample(fix=True): ceil(N * sr_out / sr_in); resampy itself produces int(N * ratio) samples and
// librosa.util.fix_length pads the (at most one) missing sample with zero
long ts_resample_kaiser_len(long N, int sr_in, int sr_out) {
    if (N < 0 || sr_in < 1 || sr_out < 1) return -1;
    return (long)std::ceil((double)N * (double)sr_out / (double)sr_in);
}
int ts_resample_kaiser(ts_ctx *ctx, const float *wav, int B, long N, int sr_in, int sr_out, float *out, void *stream) {
    if (!ctx || !wav || !out) return fail("ts_resample_kaiser: null argument");
    if (B < 1 || N < 1 || sr_in < 1 || sr_out < 1) return fail("ts_resample_kaiser: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const long Nfix = ts_resample_kaiser_len(N, sr_in, sr_out);
    MiscScope ms(ctx, s);
    if (sr_in == sr_out) {
        TS_HIP(hipMemcpyAsync(out, wav, (size_t)B * N * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    const KaiserTable *kt = nullptr;
    TS_TRY(kaiser_table(ctx->device, &kt));
    const double ratio = (double)sr_out / (double)sr_in;
    const long Nres = (long)((double)N * ratio);          // int(shape * sample_ratio)
    if (Nres < 1) return fail("ts_resample_kaiser: input too short for this rate change");
    if (Nres < Nfix) TS_HIP(hipMemsetAsync(out, 0, (size_t)B * Nfix * sizeof(float), s));
    // rows of Nfix samples; the (at most one) sample beyond Nres stays zero (fix_length)
    const hipError_t e = launch_resample_kaiser(wav, B, (int)N, kt->win.f(), kt->delta.f(), kt->nwin, kt->num_table, ratio, out,
                                                (int)Nres, (int)Nfix, s);
    TS_HIP(e);
    return 0;
}

// number of MFCC frames for N input samples: T = floor(N_resampled / hop) + 1 (center=True)
int ts_mfcc_num_frames(const ts_mfcc *m, long N) { return m ? (int)(m->resampled_len(N) / m->hop) + 1 : -1; }

// wav_dev (B,N) mono fp32 at sr_in -> feat_dev (B,T,64), T = ts_mfcc_num_frames(m, N)
int ts_mfcc_forward(ts_mfcc *m, const float *wav, int B, long N, float *feat, void *stream) {
    if (!m || !wav || !feat) return fail("ts_mfcc_forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    ts_ctx *ctx = m->ctx;
    const long N22 = m->resampled_len(N);
    if (N22 <= m->nfft / 2) return fail("ts_mfcc_forward: clip shorter than half an FFT window (reflect padding undefined)");
    const int T = (int)(N22 / m->hop) + 1;
    const long M = (long)B * T;
    ts_mfcc::Work &w = m->work(s);
    const size_t F = sizeof(float);
    const float *x22 = wav;
    {
        MiscScope ms(ctx, s);
        if (m->sr_in != m->sr_out) {
            TS_TRY(w.x22.ensure((size_t)B * N22 * F));
            TS_HIP(launch_resample_polyphase(wav, B, (int)N, m->rs_kern.f(), m->norig, m->nnew, m->width, m->kw, w.x22.f(), (int)N22, s));
            x22 = w.x22.f();
        }
        // framing + window + 2048-point real FFT + |X|^2 in one kernel (mfcc.hip::stft_power_kernel)
        TS_TRY(w.power.ensure((size_t)M * m->nbins_pad * F));
        TS_HIP(launch_stft_power(x22, B, (int)N22, T, m->hop, m->window.f(), m->tw1024.f(), m->tw2048.f(), w.power.f(), m->nbins_pad, s));
    }
    TS_TRY(w.melb.ensure((size_t)M * m->nmels * F));
    ConvParams p;
    conv_layer_params(m->mel, w.power.f(), m->nbins_pad, 1, (int)M, nullptr, 0, w.melb.f(), m->nmels, 0, m->nmels, &p);
    TS_TRY(run_conv(ctx, p, 0, s));
    {
        MiscScope ms(ctx, s);
        TS_HIP(launch_db_topdb(w.melb.f(), B, (long)T * m->nmels, 80.0f, s));
    }
    conv_layer_params(m->dct, w.melb.f(), m->nmels, 1, (int)M, nullptr, 0, feat, m->nmfcc, 0, m->nmfcc, &p);
    TS_TRY(run_conv(ctx, p, 0, s));
    return 0;
}

}  // extern "C"
