// Dev-only tool (golden-vector generation in the build container; never shipped to the GPU box,
// never part of the product). Original code, re-created from SURVEY.md Appendix E2.
// Throwaway survey tool: minimal Mach-O x86-64 user-space loader so that the bundled
// macOS RNALfold-2.1.2 (Turner-2004) can serve as an oracle on Linux. Lives in /tmp only.
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <dlfcn.h>
#include <ctype.h>
#include <errno.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>

struct mach_header_64 { uint32_t magic,cputype,cpusubtype,filetype,ncmds,sizeofcmds,flags,reserved; };
struct load_command { uint32_t cmd,cmdsize; };
struct segment_command_64 { uint32_t cmd,cmdsize; char segname[16]; uint64_t vmaddr,vmsize,fileoff,filesize; int32_t maxprot,initprot; uint32_t nsects,flags; };
struct dyld_info_command { uint32_t cmd,cmdsize,rebase_off,rebase_size,bind_off,bind_size,weak_bind_off,weak_bind_size,lazy_bind_off,lazy_bind_size,export_off,export_size; };
struct entry_point_command { uint32_t cmd,cmdsize; uint64_t entryoff,stacksize; };
#define LC_SEGMENT_64 0x19
#define LC_DYLD_INFO_ONLY 0x80000022
#define LC_MAIN 0x80000028

static uint8_t *file; static struct segment_command_64 *segs[16]; static int nsegs=0;
static long guard=0x595e9fbd94fda766L;
static void die_stub(void){ fprintf(stderr,"[mloader] unbound/unsupported symbol called\n"); abort(); }
static void my_stack_chk_fail(void){ fprintf(stderr,"[mloader] stack_chk_fail\n"); abort(); }
static int *my_error(void){ return &errno; }
static void my_memset_pattern16(void*b,const void*p,size_t len){ uint8_t*d=b; for(size_t i=0;i<len;i++) d[i]=((const uint8_t*)p)[i&15]; }
// Darwin _RuneLocale: runetype[256] at offset 60, maplower at 1084, mapupper at 2108
static uint8_t rune[4096];
static int my_maskrune(int c,unsigned long f){ if(c<0||c>255) return 0; return ((uint32_t*)(rune+60))[c]&f; }
static int my_toupper(int c){ return toupper(c); }
static int my_tolower(int c){ return tolower(c); }
static void init_rune(void){
  uint32_t *t=(uint32_t*)(rune+60); int32_t *lo=(int32_t*)(rune+60+1024), *up=(int32_t*)(rune+60+2048);
  for(int c=0;c<256;c++){ uint32_t f=0;
    if(c<128){ if(isalpha(c))f|=0x100; if(iscntrl(c))f|=0x200; if(isdigit(c))f|=0x400; if(isgraph(c))f|=0x800; if(islower(c))f|=0x1000;
      if(ispunct(c))f|=0x2000; if(isspace(c))f|=0x4000; if(isupper(c))f|=0x8000; if(isxdigit(c))f|=0x10000; if(c==' '||c=='\t')f|=0x20000; if(isprint(c))f|=0x40000; }
    t[c]=f; lo[c]=c<128?tolower(c):c; up[c]=c<128?toupper(c):c; }
}
static void *resolve(const char *name){
  if(!strcmp(name,"___stack_chk_guard")) return &guard;
  if(!strcmp(name,"___stack_chk_fail")) return (void*)my_stack_chk_fail;
  if(!strcmp(name,"___error")) return (void*)my_error;
  if(!strcmp(name,"___bzero")) return (void*)bzero;
  if(!strcmp(name,"_memset_pattern16")) return (void*)my_memset_pattern16;
  if(!strcmp(name,"__DefaultRuneLocale")) return rune;
  if(!strcmp(name,"___maskrune")) return (void*)my_maskrune;
  if(!strcmp(name,"___toupper")) return (void*)my_toupper;
  if(!strcmp(name,"___tolower")) return (void*)my_tolower;
  if(!strcmp(name,"___stdinp")) return &stdin;
  if(!strcmp(name,"___stdoutp")) return &stdout;
  if(!strcmp(name,"___stderrp")) return &stderr;
  if(!strcmp(name,"dyld_stub_binder")) return (void*)die_stub;
  void *p=dlsym(RTLD_DEFAULT,name+1);            // strip one leading underscore
  if(!p){ fprintf(stderr,"[mloader] warn: unresolved %s\n",name); return (void*)die_stub; }
  return p;
}
static uint64_t uleb(uint8_t **p){ uint64_t r=0; int s=0; uint8_t b; do{ b=*(*p)++; r|=(uint64_t)(b&0x7f)<<s; s+=7; }while(b&0x80); return r; }
static int64_t sleb(uint8_t **p){ int64_t r=0; int s=0; uint8_t b; do{ b=*(*p)++; r|=(int64_t)(b&0x7f)<<s; s+=7; }while(b&0x80); if(b&0x40) r|=-(1LL<<s); return r; }
static void do_binds(uint8_t *p,uint8_t *end,int lazy){
  const char *sym=NULL; uint64_t addr=0; int64_t addend=0; int seg=0;
  while(p<end){ uint8_t op=*p&0xF0, imm=*p&0x0F; p++;
    switch(op){
      case 0x00: if(!lazy) return; break;                 // DONE
      case 0x10: case 0x30: break;                        // SET_DYLIB_ORDINAL_IMM / SPECIAL_IMM
      case 0x20: uleb(&p); break;                         // SET_DYLIB_ORDINAL_ULEB
      case 0x40: sym=(const char*)p; p+=strlen(sym)+1; break;
      case 0x50: break;                                   // SET_TYPE_IMM
      case 0x60: addend=sleb(&p); break;
      case 0x70: seg=imm; addr=segs[seg]->vmaddr+uleb(&p); break;
      case 0x80: addr+=uleb(&p); break;
      case 0x90: *(uint64_t*)addr=(uint64_t)resolve(sym)+addend; addr+=8; break;
      case 0xA0: *(uint64_t*)addr=(uint64_t)resolve(sym)+addend; addr+=8+uleb(&p); break;
      case 0xB0: *(uint64_t*)addr=(uint64_t)resolve(sym)+addend; addr+=8+imm*8; break;
      case 0xC0: { uint64_t cnt=uleb(&p), skip=uleb(&p); for(uint64_t i=0;i<cnt;i++){ *(uint64_t*)addr=(uint64_t)resolve(sym)+addend; addr+=8+skip; } } break;
      default: fprintf(stderr,"[mloader] bad bind opcode %x\n",op); exit(2);
    } }
}
int main(int argc,char**argv,char**envp){
  if(argc<2){ fprintf(stderr,"usage: mloader macho [args]\n"); return 2; }
  int fd=open(argv[1],O_RDONLY); struct stat st; fstat(fd,&st);
  file=mmap(0,st.st_size,PROT_READ,MAP_PRIVATE,fd,0);
  struct mach_header_64 *mh=(void*)file; if(mh->magic!=0xfeedfacf){ fprintf(stderr,"not macho64\n"); return 2; }
  uint8_t *lc=file+sizeof(*mh); struct dyld_info_command *di=NULL; uint64_t entryoff=0, textbase=0;
  for(uint32_t i=0;i<mh->ncmds;i++){ struct load_command *c=(void*)lc;
    if(c->cmd==LC_SEGMENT_64){ struct segment_command_64 *s=(void*)lc; segs[nsegs++]=s;
      if(strcmp(s->segname,"__PAGEZERO") && s->vmsize){
        void *m=mmap((void*)s->vmaddr,(s->vmsize+4095)&~4095UL,PROT_READ|PROT_WRITE|PROT_EXEC,MAP_PRIVATE|MAP_ANONYMOUS|MAP_FIXED_NOREPLACE,-1,0);
        if(m!=(void*)s->vmaddr){ perror("mmap seg"); return 2; }
        memcpy(m,file+s->fileoff,s->filesize);
        if(!strcmp(s->segname,"__TEXT")) textbase=s->vmaddr; } }
    else if(c->cmd==LC_DYLD_INFO_ONLY) di=(void*)lc;
    else if(c->cmd==LC_MAIN) entryoff=((struct entry_point_command*)lc)->entryoff;
    lc+=c->cmdsize; }
  init_rune(); dlopen("libm.so.6",RTLD_NOW|RTLD_GLOBAL); dlopen("libstdc++.so.6",RTLD_NOW|RTLD_GLOBAL);
  if(di){ do_binds(file+di->bind_off,file+di->bind_off+di->bind_size,0);
          do_binds(file+di->lazy_bind_off,file+di->lazy_bind_off+di->lazy_bind_size,1); }
  int (*entry)(int,char**,char**,char**)=(void*)(textbase+entryoff);
  char *apple[]={argv[1],NULL};
  int rc=entry(argc-1,argv+1,envp,apple);
  fflush(stdout); _exit(rc);   // skip our atexit; the guest called nothing special
}
