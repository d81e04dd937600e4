#define HTS_VERSION "1.3"
